# which per-stream kernels slow down when four streams share a CU?  kernel trace of 1024 x 256 KiB against 512 x 256 KiB of text:
# a kernel whose time does not halve with half the streams is bound by a stream's own chain, one that halves by issue slots / bandwidth
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tot in 256 128; do
rm -rf gpurun_out/kt_h; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_h -- python3 tools/many_chunks_check.py $tot 256 ${1:-text} > gpurun_out/kt_h_$tot.txt 2>&1
for f in gpurun_out/kt_h/*/*kernel_stats.csv; do [ -f "$f" ] && cp "$f" gpurun_out/kt_half_${tot}_kernel_stats.csv; done
rm -rf gpurun_out/kt_h; echo "== $tot MiB"; python3 profiles/agg_kernel_stats.py gpurun_out/kt_half_${tot}_kernel_stats.csv 3 | head -24
done
