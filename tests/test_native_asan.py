"""Host-side C++ of the library under AddressSanitizer + UBSan (CPU build; the GPU pool has no ASAN)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "proqa_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"), reason="needs g++ and ROCm headers")
def test_npy_io_wordpiece_and_rand_perm_under_asan(tmp_path):
    exe = tmp_path / "npy_asan"
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "native", "npy_asan_driver.cpp"),
           os.path.join(CSRC, "npy_io.cpp"), os.path.join(CSRC, "common.cpp"), os.path.join(CSRC, "wordpiece.cpp"), "-lpthread", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    run = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, env=env)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "asan driver ok" in run.stdout
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
