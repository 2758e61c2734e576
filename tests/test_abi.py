"""CPU-side checks of the drop-in boundary: the library builds, loads, and exports exactly what include/x3hip.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from x3_compressor_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.DEFAULT_SO):
        subprocess.run(["make", "-C", os.path.join(ROOT, "x3_compressor_amd", "csrc"), "all"], check=True, capture_output=True)
    return _lib.load_library()


def declared_functions():
    txt = open(os.path.join(ROOT, "include", "x3hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(x3h_[a-z0-9_]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_lib.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_library_is_gfx950_only():
    """Every device code object bundled into the library targets gfx950 (no multi-arch / dual paths)."""
    blob = open(_lib.DEFAULT_SO, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}, targets
    assert b"nvptx" not in blob


def test_scalar_entry_points(lib):
    assert lib.x3h_abi_version() == _lib.ABI_VERSION == 6
    assert lib.x3h_strerror(0) == b"ok" and b"output" in lib.x3h_strerror(-3)
    assert lib.x3h_compress_bound(0) >= 4 and lib.x3h_compress_bound(1000) >= 2000
    p = _lib.Params()
    lib.x3h_default_params(C.byref(p))
    assert (p.window_bytes, p.max_match_count, p.factor1, p.factor2, p.nl_mode) == (8192, 15, 4, 0, 0)  # backend.c:8,21,33-34


def test_fails_loudly_without_gpu(lib):
    """No CPU fallback: without a device the handle cannot even be created."""
    if lib.x3h_device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.x3h_ctx_create(C.byref(h), 0) == -5  # X3H_E_NO_DEVICE
    with pytest.raises(_lib.X3Error):
        _lib.X3Context(0)


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load_library(str(tmp_path / "nope.so"))


def test_cli_links_only_the_c_abi():
    src = open(os.path.join(ROOT, "x3_compressor_amd", "csrc", "x3_cli.c")).read()
    assert "hip/hip_runtime" not in src and "x3hip.h" in src
    assert '"zdfkht:w:m:n:xg:"' in src  # the reference's getopt string (x3.c:484) plus the additive -g


def test_cli_without_gpu_fails_loudly(tmp_path, lib):
    """The built CLI runs here too: option parsing and file-name rules work, the hot path refuses to run without a GPU."""
    if lib.x3h_device_count() > 0:
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "x3_compressor_amd", "csrc", "x3")
    f = tmp_path / "in.txt"
    f.write_bytes(b"hello hello hello")
    r = subprocess.run([exe, "-z", str(f)], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
    r = subprocess.run([exe, "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "--chunk-kib" in r.stderr and "-w NUM" in r.stderr
