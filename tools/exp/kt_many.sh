# kernel trace of one many-chunk batch with extra environment (e.g. X3H_ARRANGE=1): bash tools/exp/kt_many.sh [mix|text]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt_m; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_m -- python3 tools/many_chunks_check.py 256 256 ${1:-mix} > gpurun_out/kt_m.txt 2>&1
for f in gpurun_out/kt_m/*/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" gpurun_out/kt_many_kernel_stats.csv; done
rm -rf gpurun_out/kt_m; python3 profiles/agg_kernel_stats.py gpurun_out/kt_many_kernel_stats.csv 3 | head -${2:-22}
