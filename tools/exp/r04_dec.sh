#!/bin/bash
# decoder round 4: GPU tests of the decoder, config-5 decode time, SQ counters per parse step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
( time timeout -k 10 700 python -m pytest tests -x -q -m gpu -k "decod or decompress or round_trip or container or damaged or capacity or migrate or cli" --durations=8 ) > gpurun_out/r04d/pytest.log 2>&1
echo "pytest rc $?" | tee gpurun_out/r04d/rc.txt
tail -n 14 gpurun_out/r04d/pytest.log
timeout -k 10 300 python3 tools/exp/cfg5_dec.py > gpurun_out/r04d/cfg5.txt 2>&1; echo "cfg5 rc $?"; cat gpurun_out/r04d/cfg5.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/r04d/pmc2 -- python3 tools/exp/dec_only.py > gpurun_out/r04d/pmc2.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d gpurun_out/r04d/pmc3 -- python3 tools/exp/dec_only.py > gpurun_out/r04d/pmc3.txt 2>&1
grep steps gpurun_out/r04d/pmc2.txt
python3 - <<PY | tee gpurun_out/r04d/step_counters.txt
import csv, glob, collections, re
steps = int(re.search(r"steps (\d+)", open("gpurun_out/r04d/pmc2.txt").read()).group(1))
print("one stream of 262144 bytes of english-like text, -w 64 -t 256:", steps, "parse steps; x3_decode_kernel (the chain), per parse step:")
for d in ("pmc2", "pmc3"):
    for f in sorted(glob.glob("gpurun_out/r04d/%s/**/*counter_collection.csv" % d, recursive=True))[-1:]:
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "x3_decode" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
        print({k: round(v / steps, 1) for k, v in acc.items()})
PY
timeout -k 10 400 python3 tools/exp/many_decode.py 2>&1 | tee gpurun_out/r04d/many_decode.txt
