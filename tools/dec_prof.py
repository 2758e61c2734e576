"""experiment: where the decoder chain's cycles go -- needs a PROFILE build of decode.hip (-DX3_DEC_PROFILE: section clocks reported through X3H_DEBUG); usage: dec_prof.py lib.so"""
import sys, os, time
sys.path.insert(0, '.')
from x3_compressor_amd import _lib, synth
ctx = _lib.X3Context(0, library=sys.argv[1] if len(sys.argv) > 1 else None)
for name, data, kw in (("mr 1 MB -w512 -t4096", synth.mr_like(1_000_000).tobytes(), dict(w_kib=512, t=4096)),
                       ("text 256 KiB -w64 -t256", synth.english_like(262144).tobytes(), dict(w_kib=64, t=256))):
    os.environ.pop("X3H_DEBUG", None)
    stream = ctx.compress(data, _lib.make_params(**kw))
    steps = ctx.last_stats.steps
    os.environ["X3H_DEBUG"] = "1"
    back = ctx.decompress(stream, len(data) + 16)
    print(name, back == data, f"steps {steps}, kernel {ctx.last_stats.ms_code:.1f} ms = {ctx.last_stats.ms_code*1e6/steps:.0f} ns/step", flush=True)
