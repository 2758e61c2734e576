#!/bin/bash
# x3_segscan_kernel alone: read and write HBM bytes (separate --pmc passes) and its phase clocks, on the many-chunk batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/sg_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/sg_$c -- python3 tools/many_chunks_check.py 256 256 mix > gpurun_out/sg_$c.txt 2>&1 || exit 1
done
python3 - <<'P'
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/sg_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "seg" in k or "walk" in k: acc[k].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(c, k, "launches", len(v), "last: %.2f GB (KB units x 1024; FETCH not yet doubled)" % (v[-1] * 1024 / 1e9))
P
rm -rf gpurun_out/sg_FETCH_SIZE gpurun_out/sg_WRITE_SIZE
X3H_SEG_PROF=1 python3 tools/many_chunks_check.py 256 256 mix 2>&1 | grep -a "scan3\|MB/s" | tail -4
