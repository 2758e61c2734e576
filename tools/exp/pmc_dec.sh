# experiment: per-parse-step SQ counters of one single-stream decode (x3_decode_kernel, one wave): bash tools/exp/pmc_dec.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/decpmc2 -- python3 tools/exp/dec_only.py > gpurun_out/decpmc2.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d gpurun_out/decpmc3 -- python3 tools/exp/dec_only.py > gpurun_out/decpmc3.txt 2>&1
grep steps gpurun_out/decpmc2.txt
python3 - <<PY
import csv, glob, collections, re
steps = int(re.search(r"steps (\d+)", open("gpurun_out/decpmc2.txt").read()).group(1))
for d in ("decpmc2", "decpmc3"):
    for f in sorted(glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True))[-1:]:
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "decode" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
        print({k: round(v / steps, 1) for k, v in acc.items()}, "per step")
PY
