#!/bin/bash
# the whole GPU suite as the driver runs it, then the bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
( time timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 ) > gpurun_out/r04g/pytest_gpu.log 2>&1
echo "pytest rc $?" | tee gpurun_out/r04g/rc.txt
tail -n 30 gpurun_out/r04g/pytest_gpu.log
( time timeout -k 10 900 python bench.py ) > gpurun_out/r04g/bench.json 2> gpurun_out/r04g/bench.err
echo "bench rc $?" | tee -a gpurun_out/r04g/rc.txt
tail -n 5 gpurun_out/r04g/bench.err
python3 - <<'P'
import json
try:
    l = json.loads(open("gpurun_out/r04g/bench.json").read().strip().splitlines()[0])
    print({k: l[k] for k in ("value", "ms_per_step", "ratio", "stream_sha256_equals_reference")}, l["stage_ms"], l["roofline"]["frac"], l["roofline"]["frac_path"])
    print("chunked:", [(e["chunks"], e["value"], e["ratio"]) for e in l["chunked_same_bytes"]["sweep"]], l["chunked_same_bytes"]["best_ratio_at_1GBps"])
    print("many:", {k: l["many_chunks_batch"][k] for k in ("value", "ms", "ratio", "stage_ms")})
    print("cfg4:", l["config4_share_per_gpu"]); print("cfg3:", l.get("config3_full")); print("cfg5:", l.get("config5_round_trip")); print("cpu:", l.get("cpu_baseline"))
except Exception as e:
    print("bench parse failed", e)
P
