"""Properties of the COMPILED decoder chain that its speed depends on and that a source edit can silently lose (decode.hip, the notes at dec_loop and
dec_request).  hipcc cross-compiles gfx950 without a GPU; the checks read the generated ISA."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "x3_compressor_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def decode_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path_factory.mktemp("isa") / "decode.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S", "-o", str(out), "decode.hip"],
                   cwd=CSRC, check=True, capture_output=True, timeout=600)
    text = open(out).read()
    kernels = {}
    for name in ("x3_decode_kernel", "x3_decode_mid_kernel", "x3_decode_eight_kernel", "x3_decode_many_kernel"):
        m = re.search(r"^_Z\d+%s9X3DecArgs:[^\n]*\n(.*?)^\s*\.amdhsa_kernel" % name, text, re.S | re.M)
        assert m, name
        kernels[name] = [l.strip() for l in m.group(1).splitlines() if l.strip() and not l.strip().startswith(";")]
    return kernels


def test_chain_state_stays_on_the_scalar_unit(decode_isa):
    """The chain's state is wave-uniform.  If the compiler's uniformity analysis gives up on the main loop (a lane-dependent branch next to one of its failing
    exits makes it a 'cycle with divergent exit'), every scalar of the state becomes a vector register and every compare a v_cmp: the step takes twice as long.
    Measured on the two builds: scalar compares 206 / vector compares 147 (good), 70 / 325 (bad)."""
    for name, lines in decode_isa.items():
        s_cmp = sum(1 for l in lines if l.startswith("s_cmp_"))
        v_cmp = sum(1 for l in lines if l.startswith("v_cmp"))
        assert s_cmp > v_cmp, f"{name}: {s_cmp} scalar compares, {v_cmp} vector compares -- the chain's state has gone to the vector unit"


def test_requested_blocks_are_touched_by_nothing_but_request_and_take(decode_isa):
    """dec_request / dec_take (decode.hip): the context blocks of the next step are loaded into v120..v124 by two asm loads and taken over behind one s_waitcnt; no
    other instruction of the kernel may name those registers (a compiler temporary there would race with the loads in flight)."""
    for name, lines in decode_isa.items():
        uses = [l for l in lines if re.search(r"\bv12[0-4]\b|v\[12[0-4]:12[0-4]\]", l)]
        assert uses, name
        for l in uses:
            ok = (re.match(r"global_load_dwordx3 v\[122:124\], v\d+, s\[\d+:\d+\]$", l) or re.match(r"global_load_dwordx2 v\[120:121\], v\d+, s\[\d+:\d+\]$", l)
                  or re.match(r"v_mov_b32 v\d+, v12[0-4]$", l))
            assert ok, f"{name}: unexpected use of the request registers: {l}"
        nv = max(int(x) for l in lines for x in re.findall(r"\bv(\d+)\b", l))
        assert nv <= 127, f"{name}: v{nv} in use -- more than 128 vector registers, fewer than four wavefronts per SIMD (the many-stream variant runs sixteen streams per CU)"
        takes = [i for i, l in enumerate(lines) if re.match(r"v_mov_b32 v\d+, v120$", l)]
        assert takes and all(lines[i - 1] == "s_waitcnt vmcnt(0)" for i in takes), f"{name}: a take without its wait"
