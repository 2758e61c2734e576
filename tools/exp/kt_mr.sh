cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt_m; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_m -- python3 tools/many_chunks_check.py 64 256 mr > gpurun_out/kt_m.txt 2>&1
for f in gpurun_out/kt_m/*/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" gpurun_out/kt_mr_kernel_stats.csv; done
rm -rf gpurun_out/kt_m; python3 profiles/agg_kernel_stats.py gpurun_out/kt_mr_kernel_stats.csv 3 | head -12
