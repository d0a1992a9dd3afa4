#!/bin/bash
# Round-4 GPU session A: box facts, GPU tests, the dense-writer control experiment (+ its PMC pass), one bench line.
#   gpurun --timeout 1190 -- 'bash tools/gpu_r04a.sh r04a'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r04a}
mkdir -p $O
{ rocm-smi --showmemorypartition --showcomputepartition --showclocks --showpower --showmeminfo vram 2>&1; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" ; } > $O/${TAG}_box.txt 2>&1
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/${TAG}_pytest.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 400 python tools/dense_control.py > $O/${TAG}_dense_control.jsonl 2> $O/${TAG}_dense_control.err; rc=$?; echo "dense_control rc=$rc"
[ $rc -eq 124 ] && exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/${TAG}_counters.txt 2>&1
grep -o "TCC_EA0_WRREQ[A-Za-z0-9_]*" $O/${TAG}_counters.txt | sort -u | tr '\n' ' '; echo
if grep -q "TCC_EA0_WRREQ_STALL" $O/${TAG}_counters.txt; then
  timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum --kernel-trace --output-format csv -d $O/${TAG}_pmc_wrreq -- python3 $R/tools/dense_control.py --pmc --buffers 3 > $O/${TAG}_pmc_wrreq.log 2>&1; echo "pmc wrreq rc=$?"
fi
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -- python3 $R/tools/dense_control.py --pmc --buffers 3 > $O/${TAG}_pmc_write.log 2>&1; echo "pmc write rc=$?"
cd $R
timeout -k 10 400 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
ls $O | grep ${TAG} | wc -l
