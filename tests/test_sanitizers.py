"""the host sanitizer target.  `make -C tests/emu asan` compiles the kernel + host sources (SIMT emulator build)
and the oracle under -fsanitize=address,undefined and runs them through the C ABI on small inputs (tests/emu/asan_selftest.cpp).
GPU AddressSanitizer is not available on the pool, so this CPU build is where memory errors in the kernels' index logic show up."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_asan_ubsan_selftest():
    r = subprocess.run(["make", "-j8", "-C", os.path.join(HERE, "emu"), "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "asan_selftest ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
