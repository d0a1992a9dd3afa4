#!/bin/bash
# One GPU session: tests, bench, rocprof stats + PMC passes.  Run via gpurun from the repo root.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r01}
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/${TAG}_pytest.log
tail -3 $O/${TAG}_pytest.log
python bench.py --extras > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
cat $O/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_stats -- python3 $R/bench.py --no-cpu --steps 200 --warmup 20 > $O/${TAG}_prof_stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_prof_fetch -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 > $O/${TAG}_prof_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof_write -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 > $O/${TAG}_prof_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_prof_sq -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 > $O/${TAG}_prof_sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_micro -- python3 $R/tools/microbench.py adi expand dense code > $O/${TAG}_prof_micro.log 2>&1; echo "micro stats rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof_microwrite -- python3 $R/tools/microbench.py adi expand > $O/${TAG}_prof_microwrite.log 2>&1; echo "micro write rc=$?"
find $O -name "*.csv" | grep ${TAG} | head -30
