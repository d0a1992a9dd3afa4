#!/bin/bash
# Round-6 sessions.  part 1: GPU tests, smoke, bench lines (default and the driver's 20-step shape: compact line + full record), rocprofv3
#                            kernel stats (headline alone, then with the configs)
#                    part 2: PMC passes (FETCH_SIZE / WRITE_SIZE on the headline, WRITE_SIZE + SQ with the configs), tool benches
#   gpurun --timeout 1190 -- 'bash tools/gpu_r06.sh r06 1'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r06}
PART=${2:-1}
mkdir -p $O
cd $R
if [ "$PART" = "1" ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/${TAG}_pytest.log
  tail -3 $O/${TAG}_pytest.log
  python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/${TAG}_smoke.log
  python bench.py --full-out $O/${TAG}_bench_full.json > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
  ( time python bench.py --gpus 1 --steps 20 --warmup 5 --full-out $O/${TAG}_bench20_full.json > $O/${TAG}_bench20.json ) 2> $O/${TAG}_bench20.err; echo "bench (driver shape: 20 steps) rc=$?"
  wc -c $O/${TAG}_bench.json $O/${TAG}_bench20.json; tail -4 $O/${TAG}_bench20.err
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_stats -- python3 $R/bench.py --no-cpu --no-configs --steps 200 --warmup 20 --full-out /tmp/f1.json > $O/${TAG}_prof_stats.log 2>&1; echo "stats rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_statscfg -- python3 $R/bench.py --no-cpu --steps 200 --warmup 20 --full-out /tmp/f2.json > $O/${TAG}_prof_statscfg.log 2>&1; echo "stats (configs) rc=$?"
  find $O/${TAG}_prof_stats $O/${TAG}_prof_statscfg -name "*kernel_trace.csv" -size +8M -delete
else
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_prof_fetch -- python3 $R/bench.py --no-cpu --no-configs --steps 10 --warmup 2 --full-out /tmp/f3.json > $O/${TAG}_prof_fetch.log 2>&1; echo "fetch rc=$?"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof_write -- python3 $R/bench.py --no-cpu --no-configs --steps 10 --warmup 2 --full-out /tmp/f4.json > $O/${TAG}_prof_write.log 2>&1; echo "write rc=$?"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof_writecfg -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 --full-out /tmp/f5.json > $O/${TAG}_prof_writecfg.log 2>&1; echo "write (configs) rc=$?"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_prof_sq -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 --full-out /tmp/f6.json > $O/${TAG}_prof_sq.log 2>&1; echo "sq rc=$?"
  python3 $R/tools/pmc_aggregate.py $O/${TAG}_prof_fetch $O/${TAG}_prof_write $O/${TAG}_prof_writecfg $O/${TAG}_prof_sq
  find $O -name "*kernel_trace.csv" -path "*${TAG}_prof*" -delete
  du -sh $O
  cd $R
  python tools/bench_cfg5.py > $O/${TAG}_cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
  python tools/bench_adi_pipeline.py > $O/${TAG}_adi_pipeline.json 2>/dev/null; echo "adi pipeline rc=$?"
  python tools/bench_adi_pipeline.py 200 20000 --depth 14 --cube-size 2 > $O/${TAG}_adi_pipeline_222.json 2>/dev/null; echo "adi pipeline 2x2x2 rc=$?"
fi
