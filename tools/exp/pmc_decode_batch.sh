# experiment: HBM traffic of the decode kernel for the 1024 x 256 KiB text batch (separate --pmc passes).  bash tools/exp/pmc_decode_batch.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pd_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pd_$c -- python3 tools/decode_batch_check.py 256 256 > gpurun_out/pd_$c.txt 2>&1
  echo "$c pass done"
done
python3 - <<PY
import csv, glob, json
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, n = 0.0, 0
    for f in glob.glob("gpurun_out/pd_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "x3_decode" in r["Kernel_Name"] and r["Counter_Name"] == c:
                tot += float(r["Counter_Value"]); n += 1
    res[c] = (tot, n)
launches = max(res["FETCH_SIZE"][1], 1)
fetch = 2 * res["FETCH_SIZE"][0] * 1024 / launches      # KB -> B, doubled on gfx950 (MI355X_MICROARCH.md)
write = res["WRITE_SIZE"][0] * 1024 / max(res["WRITE_SIZE"][1], 1)
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 tools/decode_batch_check.py 256 256 (two passes)", "kernel": "x3_decode_mid_kernel", "launches_seen": launches,
       "batch_bytes": 256 << 20, "hbm_read_GB_per_launch": round(fetch / 1e9, 2), "hbm_written_GB_per_launch": round(write / 1e9, 2),
       "bytes_per_decoded_byte": round((fetch + write) / (256 << 20), 1)}
print(json.dumps(out)); json.dump(out, open("gpurun_out/r02_decode_pmc_traffic.json", "w"), indent=1)
PY
