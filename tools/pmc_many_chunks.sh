# HBM traffic of one many-chunk batch by kernel family: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/many_chunks_check.py,
# aggregated by tools/pmc_traffic.py (last batch of the run; FETCH doubled per MI355X_MICROARCH.md).  usage: bash tools/pmc_many_chunks.sh [mix|text] [out.json]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KIND=${1:-mix}; OUT=${2:-gpurun_out/r03_many_chunks_pmc_traffic.json}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pm_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pm_$c -- python3 tools/many_chunks_check.py 256 256 $KIND > gpurun_out/pm_$c.txt 2>&1 || exit 1
  echo "$c pass done"
done
F=$(find gpurun_out/pm_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find gpurun_out/pm_WRITE_SIZE -name '*counter_collection.csv' | head -1)
PMC_SPLIT=1 python3 tools/pmc_traffic.py batch $F $W ${OUT%.json}_by_pass.json 268435456 > /dev/null
python3 tools/pmc_traffic.py batch $F $W $OUT 268435456 && python3 -c "
import json; t=json.load(open('$OUT')); print(json.dumps(t['total'])); [print(f'{k:44s}', v) for k,v in list(t['groups'].items())[:14]]"
rm -rf gpurun_out/pm_FETCH_SIZE gpurun_out/pm_WRITE_SIZE
