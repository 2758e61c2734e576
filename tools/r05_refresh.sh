# round-5 measurement refresh on the GPU box (everything the DESIGN.md tables and bench.py's PMC-derived fields cite):
#   bash tools/r05_refresh.sh        (about 6 GPU-minutes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r05
# 1. kernel trace of the bench step (single stream) and of the many-chunk batch
rm -rf gpurun_out/kt; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --no-cpu --no-secondary --steps 4 --warmup 1 > gpurun_out/${R}_final_bench_under_rocprof.json 2> gpurun_out/kt.err && echo kt-ok
for f in gpurun_out/kt/*/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" gpurun_out/${R}_final_kernel_stats.csv; done; rm -rf gpurun_out/kt
rm -rf gpurun_out/km; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/km -- python3 tools/many_chunks_check.py 256 256 mix > gpurun_out/${R}_many_chunks_run.txt 2> gpurun_out/km.err && echo km-ok
for f in gpurun_out/km/*/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" gpurun_out/${R}_many_chunks_mix_kernel_stats.csv; done; rm -rf gpurun_out/km
# 2. HBM traffic (separate --pmc passes): the bench step by kernel, the many-chunk batch by kernel family
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pb_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pb_$c -- python3 bench.py --no-cpu --no-secondary --steps 2 --warmup 1 > gpurun_out/pb_$c.txt 2>&1 || exit 1
done
F=$(find gpurun_out/pb_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find gpurun_out/pb_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py bench $F $W gpurun_out/${R}_pmc_traffic.json 10192446 64 256 3 > /dev/null && echo pmc-bench-ok
rm -rf gpurun_out/pb_FETCH_SIZE gpurun_out/pb_WRITE_SIZE
bash tools/pmc_many_chunks.sh mix gpurun_out/${R}_many_chunks_pmc_traffic.json > gpurun_out/pmc_many.log 2>&1 && echo pmc-many-ok
# 3. the N > 1 leg rehearsed with one rank (config 4 weak form + the framed gather), then the driver-format line
X3_BENCH_LEG=config4 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 > gpurun_out/${R}_bench_config4_leg_one_rank.json 2> gpurun_out/c4.err && echo c4-ok
python3 bench.py > gpurun_out/${R}_bench_final.json 2> gpurun_out/bench_final.err && echo bench-ok
# 4. the decoder's chain: SQ counters per parse step of one stream, and the 1024-stream batch
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/dp2 -- python3 tools/dec_only.py > gpurun_out/dp2.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d gpurun_out/dp3 -- python3 tools/dec_only.py > gpurun_out/dp3.txt 2>&1
python3 - <<PY > gpurun_out/${R}_dec_step_counters.txt
import csv, glob, collections, re
steps = int(re.search(r"steps (\d+)", open("gpurun_out/dp2.txt").read()).group(1))
print("x3_decode_kernel (stage 1 of the decoder: the chain, ONE wavefront), one stream of 262144 bytes of english-like text, -w 64 -t 256 =", steps, "parse steps")
print("rocprofv3 --pmc, two separate passes of tools/dec_only.py; every value divided by the number of parse steps; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count in units of 4 cycles")
print("round 3 (profiles of DESIGN.md section 8): 607 instructions per step = 343 scalar + 180 vector + 65 branches + 10 LDS + 9 global; 4 300 cycles per step")
for d in ("dp2", "dp3"):
    for f in sorted(glob.glob("gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True))[-1:]:
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "x3_decode" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
        print({k: round(v / steps, 1) for k, v in sorted(acc.items())})
        if d == "dp3":
            tot = sum(acc[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")) / steps
            print("instructions per parse step (VALU + SALU + branch + LDS + VMEM):", round(tot, 1))
PY
rm -rf gpurun_out/dp2 gpurun_out/dp3
python3 tools/decode_single_streams.py > gpurun_out/${R}_decode_single_streams.txt 2>&1
python3 tools/decode_batch_1024.py > gpurun_out/${R}_decode_batch_1024.txt 2>&1
echo refresh-done
