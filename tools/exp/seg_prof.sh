# phase cycle counters of the per-chunk K1 (scan3.hip) + the kernel trace of one many-chunk batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
X3H_SEG_PROF=1 python tools/many_chunks_check.py 256 256 ${1:-mix} > gpurun_out/seg_prof.txt 2>&1
rm -rf gpurun_out/kt_seg; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_seg -- python3 tools/many_chunks_check.py 256 256 ${1:-mix} > gpurun_out/seg_kt.txt 2>&1
for f in gpurun_out/kt_seg/*/*kernel_stats.csv; do [ -f "$f" ] && cut -c1-100,200-400 "$f" | head -30 > gpurun_out/seg_kernel_stats.txt && cp "$f" gpurun_out/seg_kernel_stats.csv; done
rm -rf gpurun_out/kt_seg
cat gpurun_out/seg_prof.txt
