"""The C-ABI library loads, exports every symbol include/proqa_hip.h declares, and its .npy
reader/writer is byte-compatible with numpy (no GPU needed for either)."""
import os
import re

import numpy as np
import pytest

from proqa_amd import _lib, npy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(_lib.LIB_PATH):
        from proqa_amd import build
        build.build()


def declared_functions():
    text = open(os.path.join(ROOT, "include", "proqa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(proqa_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in proqa_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.proqa_abi_version() == 7


def test_error_reporting_without_compute():
    lib = _lib.load()
    info = _lib.NpyInfo()
    rc = lib.proqa_npy_stat(b"/nonexistent/file.npy", info)
    assert rc == -4 and b"cannot open" in lib.proqa_last_error()
    with pytest.raises(_lib.ProqaError):
        _lib.check(rc)


def test_encoder_abi_rejects_bad_models_without_a_gpu():
    """proqa_encoder_create validates the weight table before touching the device."""
    import ctypes
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.proqa_encoder_create(None, ctypes.byref(h)) == -1 and b"NULL argument" in lib.proqa_last_error()
    bw = _lib.BertWeights(hidden=100, n_layers=1, n_heads=2, intermediate=64, max_position=8, vocab=10,
                          layer_norm_eps=1e-12)
    assert lib.proqa_encoder_create(ctypes.byref(bw), ctypes.byref(h)) == -1
    assert b"head_dim 64" in lib.proqa_last_error()
    bw.hidden = 128
    assert lib.proqa_encoder_create(ctypes.byref(bw), ctypes.byref(h)) == -1      # no layer table / NULL weights
    assert not h.value
    assert lib.proqa_encoder_free(None) == 0


def test_npy_reader_matches_numpy_files():
    for name, dtype in [("npy_f2.npy", np.float16), ("npy_f4.npy", np.float32)]:
        path = os.path.join(GOLDEN, name)
        want = np.load(path)
        info = npy.stat(path)
        assert (info["rows"], info["cols"], info["dtype"]) == (want.shape[0], want.shape[1], dtype)
        assert info["data_offset"] == 128
        np.testing.assert_array_equal(npy.load(path), want)
        np.testing.assert_array_equal(npy.load(path, 1, 2), want[1:3])
        np.testing.assert_array_equal(np.asarray(npy.memmap(path)), want)


def test_npy_writer_is_byte_identical_to_numpy(tmp_path):
    for name in ("npy_f2.npy", "npy_f4.npy"):
        src = os.path.join(GOLDEN, name)
        arr = np.load(src)
        out = npy.save(str(tmp_path / name.replace(".npy", "")), arr)   # '.npy' appended like np.save
        assert out.endswith(name)
        assert open(out, "rb").read() == open(src, "rb").read()
    # shapes whose header crosses a 64-byte boundary
    for rows in (0, 1, 123456789, 18_000_000):
        path = str(tmp_path / f"h{rows}.npy")
        npy.create(path, rows if rows < 1000 else 3, 128, np.float16)
    big = np.zeros((3, 128), np.float16)
    ref = str(tmp_path / "ref.npy")
    np.save(ref, big)
    assert open(str(tmp_path / "h123456789.npy"), "rb").read() == open(ref, "rb").read()


def test_npy_create_and_write_rows(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((10, 128)).astype(np.float16)
    path = npy.create(str(tmp_path / "parts"), 10, 128, np.float16)
    npy.write_rows(path, 6, a[6:])
    npy.write_rows(path, 0, a[:6])
    np.testing.assert_array_equal(np.load(path), a)
    with pytest.raises(_lib.ProqaError):
        npy.write_rows(path, 8, a[:5])
    # rows of another dtype or width must not be written into the file (they would corrupt it silently)
    with pytest.raises(_lib.ProqaError):
        npy.write_rows(path, 0, a[:2].astype(np.float32 if a.dtype == np.float16 else np.float16))
    with pytest.raises(_lib.ProqaError):
        npy.write_rows(path, 0, a[:2, :-1])


def test_npy_rejects_bad_files(tmp_path):
    p = tmp_path / "bad.npy"
    p.write_bytes(b"not a numpy file at all")
    with pytest.raises(_lib.ProqaError):
        npy.stat(str(p))
    q = tmp_path / "i8.npy"
    np.save(q, np.zeros((2, 128), np.int64))
    with pytest.raises(_lib.ProqaError):
        npy.stat(str(q))
    r = tmp_path / "f.npy"
    np.save(r, np.asfortranarray(np.zeros((2, 128), np.float32)))
    with pytest.raises(_lib.ProqaError):
        npy.stat(str(r))
    t = tmp_path / "trunc.npy"
    np.save(t, np.zeros((4, 128), np.float16))
    t.write_bytes(t.read_bytes()[:-10])
    with pytest.raises(_lib.ProqaError):
        npy.stat(str(t))


def test_npy_round_trips_fuzz(tmp_path):
    """Random shapes around the header-length boundaries (row counts with 1..9 digits change the header
    padding) and both dtypes: our writer's files are byte-identical to np.save's, our reader returns
    np.load's array, partial reads and sharded writes line up."""
    import io
    from proqa_amd import npy
    rng = np.random.default_rng(0)
    rows_choices = [0, 1, 9, 10, 99, 100, 999, 1000, 9999, 10000, 12345]
    for case in range(40):
        rows = int(rng.choice(rows_choices))
        cols = int(rng.choice([1, 7, 128]))
        dtype = np.float16 if case % 2 else np.float32
        arr = rng.standard_normal((rows, cols)).astype(dtype)
        p = str(tmp_path / f"a{case}.npy")
        npy.save(p, arr)
        buf = io.BytesIO()
        np.save(buf, arr)
        assert open(p, "rb").read() == buf.getvalue(), (rows, cols, dtype)
        np.testing.assert_array_equal(npy.load(p), arr)
        if rows > 3:
            r0 = int(rng.integers(0, rows - 1))
            nr = int(rng.integers(1, rows - r0 + 1))
            np.testing.assert_array_equal(npy.load(p, r0, nr), arr[r0:r0 + nr])
            p2 = str(tmp_path / f"b{case}.npy")
            npy.create(p2, rows, cols, dtype)
            cut = int(rng.integers(1, rows))
            npy.write_rows(p2, cut, arr[cut:])
            npy.write_rows(p2, 0, arr[:cut])
            assert open(p2, "rb").read() == buf.getvalue()


def test_driver_build_entry_point_runs():
    """__graft_entry__.build() is what the driver calls: it must compile (a no-op when up to date), load and bind
    the library and import the package."""
    import __graft_entry__
    assert os.path.exists(__graft_entry__.build())
