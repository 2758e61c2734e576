#!/bin/bash
# SQ counters of the dispatches of x3_segsort_kernel (by context1, by pair) on the many-chunk batch: what the slower one waits for
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { rm -rf gpurun_out/sp; rocprofv3 --pmc $1 --kernel-trace --output-format csv -d gpurun_out/sp -- python3 tools/many_chunks_check.py 256 256 mix > gpurun_out/sp.txt 2>&1 || { tail -3 gpurun_out/sp.txt; return; }
python3 - <<'P'
import csv, glob, collections
f = glob.glob("gpurun_out/sp/**/*counter_collection.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "segsort" in r["Kernel_Name"]: acc.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for d, v in list(acc.items())[-2:]: print(d, {k: "%.3g" % x for k, x in v.items()})
P
}
run "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
run "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE"
run "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_ATOMIC_RETURN SQ_BUSY_CYCLES"
rm -rf gpurun_out/sp
