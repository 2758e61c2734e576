"""The x3 CLI end to end on a large file: x3 -z --chunk-kib K (X3C1 container), x3 -d, cmp.  usage: cli_check.py [MiB] [chunk_KiB ...]
Prints the library-call time the CLI reports ("elapsed time", H2D/D2H included) and the process wall time (file I/O, context creation)."""
import os, re, subprocess, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from x3_compressor_amd import synth
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kibs = [int(a) for a in sys.argv[2:]] or [256, 64]
X3 = os.path.join("x3_compressor_amd", "csrc", "x3")
base = synth.english_like(8 << 20)
data = np.tile(base, max((mib << 20) // base.size, 1))[:mib << 20]
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    src = os.path.join(d, "in.bin"); data.tofile(src)
    for kib in kibs:
        comp, back = os.path.join(d, f"c{kib}.x3c"), os.path.join(d, f"b{kib}.bin")
        time.sleep(4)  # the driver clears the previous process's VRAM (tens of GB of workspace) in the background: an allocation right behind it waits for that
        t0 = time.time(); r = subprocess.run([X3, "-z", "-w", "64", "-t", "256", "--chunk-kib", str(kib), src, comp], capture_output=True, text=True); tz = time.time() - t0
        assert r.returncode == 0, r.stderr
        ez = float(re.search(r"elapsed time: ([0-9.]+)", r.stderr).group(1))
        time.sleep(4)
        t0 = time.time(); r = subprocess.run([X3, "-d", comp, back], capture_output=True, text=True); td = time.time() - t0
        assert r.returncode == 0, r.stderr
        ed = float(re.search(r"elapsed time: ([0-9.]+)", r.stderr).group(1))
        same = subprocess.run(["cmp", src, back]).returncode == 0
        n = data.size
        print(f"{mib} MiB, --chunk-kib {kib}: x3 -z library call {ez*1e3:.0f} ms = {n/ez/1e6:.0f} MB/s (process {tz:.2f} s), ratio {n/os.path.getsize(comp):.3f}; "
              f"x3 -d library call {ed*1e3:.0f} ms = {n/ed/1e6:.0f} MB/s (process {td:.2f} s); cmp clean: {same}", flush=True)
