"""experiment: ONE decode of one text stream (for rocprofv3 --pmc SQ_* : instructions per parse step by kind); prints the step count.
usage: dec_only.py [bytes]   e.g.  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d out -- python3 tools/dec_only.py"""
import sys
sys.path.insert(0, '.')
from x3_compressor_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
data = synth.english_like(n).tobytes()
ctx = _lib.X3Context(0)
stream = ctx.compress(data, _lib.make_params(w_kib=64, t=256))
steps = ctx.last_stats.steps
back = ctx.decompress(stream, n + 16)
print("steps", steps, "ok", back == data, "decode kernel ms", ctx.last_stats.ms_code)
