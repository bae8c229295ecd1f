"""Build libproqa_hip.so in-tree with hipcc for gfx950.

The library is linked WITHOUT a DT_NEEDED entry on libamdhip64: the process that loads it
decides which HIP runtime is live (PyTorch-ROCm ships its own copy; loading a second one next
to it would give the kernels a different runtime than the tensors they are handed).
proqa_amd._lib makes the runtime's symbols global before dlopen()ing the library.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libproqa_hip.so")
ARCH = "gfx950"

SOURCES = [
    "common.cpp",
    "npy_io.cpp",
    "wordpiece.cpp",
    "mips_index.cpp",
    "mips_kernels.hip",
    "sharded_search.cpp",
    "encoder_kernels.hip",
    "gemm_kernels.hip",
    "attention_kernel.hip",
    "lt_gemm.cpp",
    "encoder.cpp",
    "kmeans_kernels.hip",
    "microbench.hip",
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libproqa_hip.so cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _child_env():
    """Environment of the compiler children: a profiler's or sanitizer's LD_PRELOAD belongs to the process under
    test, not to hipcc / g++."""
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    return env


def build(force=False, verbose=False):
    """Compile every source for gfx950 and link csrc/libproqa_hip.so. Returns its path.

    Safe when several processes start at once (torchrun ranks on a never-built checkout): one exclusive lock
    around the whole build, objects and the library are written under temporary names and renamed into place,
    so nobody can dlopen a half-written file."""
    import fcntl
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "proqa_hip.h"))
    env = _child_env()
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)          # released when the file is closed
        objs = []
        todo = []
        for src in SOURCES:
            path = os.path.join(CSRC, src)
            obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
            objs.append(obj)
            if force or _stale(obj, [path] + headers):   # re-checked under the lock: another rank may have built it
                todo.append((path, obj))

        def compile_one(job):
            path, obj = job
            tmp = f"{obj}.{os.getpid()}.tmp"
            cmd = [hipcc, "-x", "hip", f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
                   "-Wall", "-Wno-unused-function", "-c", path, "-o", tmp]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            try:
                subprocess.run(cmd, check=True, env=env)
                os.replace(tmp, obj)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)

        if todo:
            # the sources are independent: a few compilers side by side (hipcc peaks at ~2 GB each)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(4, len(todo), os.cpu_count() or 1)) as pool:
                list(pool.map(compile_one, todo))        # re-raises the first CalledProcessError
        if force or _stale(LIB_PATH, objs):
            tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
            cmd = ["g++", "-shared", "-o", tmp] + objs + ["-Wl,--no-as-needed", "-lpthread", "-lm", "-ldl"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            try:
                subprocess.run(cmd, check=True, env=env)
                os.replace(tmp, LIB_PATH)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
