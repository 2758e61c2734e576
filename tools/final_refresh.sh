cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r02_bench_final.json 2> gpurun_out/bench_final.err && echo bench-ok
rm -rf gpurun_out/kt; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --no-cpu --no-secondary --steps 4 --warmup 1 > gpurun_out/r02_final_bench_under_rocprof.json 2> gpurun_out/kt.err && echo kt-ok
for f in gpurun_out/kt/*/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" gpurun_out/r02_final_kernel_stats.csv; done
rm -rf gpurun_out/kd; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kd -- python3 tools/decode_batch_check.py 256 256 > gpurun_out/r02_decode_under_rocprof.txt 2> gpurun_out/kd.err && echo kd-ok
for f in gpurun_out/kd/*/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" gpurun_out/r02_decode_kernel_stats.csv; done
grep -c . gpurun_out/r02_final_kernel_stats.csv gpurun_out/r02_decode_kernel_stats.csv
