#!/bin/bash
# Kernel-trace split of the ADI pipeline (tools/bench_adi_pipeline.py) at the three sizes, one process each.
#   gpurun --timeout 900 -- 'bash tools/gpu_adi_split.sh r05base [--graph]'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r05}
shift
mkdir -p $O
python $R/tools/bench_adi_pipeline.py "$@" > $O/${TAG}_adi_pipeline.json 2> $O/${TAG}_adi_pipeline.err; echo "pipeline rc=$?"
cat $O/${TAG}_adi_pipeline.json
cd /tmp && export TMPDIR=/tmp
for W in 200 20000 100000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_adi_trace_$W -- python3 $R/tools/bench_adi_pipeline.py $W "$@" > $O/${TAG}_adi_trace_$W.log 2>&1; echo "trace $W rc=$?"
  python3 $R/tools/adi_split.py $O/${TAG}_adi_trace_$W "${TAG} ${W}x30 $*" > $O/${TAG}_adi_split_$W.json; cat $O/${TAG}_adi_split_$W.json
  # keep the merge-back small: the stats CSV is what gets committed, the raw trace stays on the box
  find $O/${TAG}_adi_trace_$W -name "*kernel_trace.csv" -size +8M -delete
done
