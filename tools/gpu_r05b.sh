#!/bin/bash
# Round-5 final sessions.  part 1: GPU tests, bench line, rocprofv3 kernel stats (headline alone, then with the configs)
#                          part 2: PMC passes (FETCH_SIZE / WRITE_SIZE / SQ), family front writer traffic, fresh-process ADI repeats, tool benches
#   gpurun --timeout 1190 -- 'bash tools/gpu_r05b.sh r05 1'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r05}
PART=${2:-1}
mkdir -p $O
cd $R
if [ "$PART" = "1" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/${TAG}_pytest.log
  tail -3 $O/${TAG}_pytest.log
  python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/${TAG}_smoke.log
  python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
  python bench.py --steps 20 --warmup 3 > $O/${TAG}_bench20.json 2> $O/${TAG}_bench20.err; echo "bench (driver shape: 20 steps) rc=$?"
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_stats -- python3 $R/bench.py --no-cpu --no-configs --steps 200 --warmup 20 > $O/${TAG}_prof_stats.log 2>&1; echo "stats rc=$?"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_statscfg -- python3 $R/bench.py --no-cpu --steps 200 --warmup 20 > $O/${TAG}_prof_statscfg.log 2>&1; echo "stats (configs) rc=$?"
  find $O/${TAG}_prof_stats $O/${TAG}_prof_statscfg -name "*kernel_trace.csv" -size +8M -delete
else
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_prof_fetch -- python3 $R/bench.py --no-cpu --no-configs --steps 10 --warmup 2 > $O/${TAG}_prof_fetch.log 2>&1; echo "fetch rc=$?"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof_write -- python3 $R/bench.py --no-cpu --no-configs --steps 10 --warmup 2 > $O/${TAG}_prof_write.log 2>&1; echo "write rc=$?"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof_writecfg -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 > $O/${TAG}_prof_writecfg.log 2>&1; echo "write (configs) rc=$?"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_prof_fetchcfg -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 > $O/${TAG}_prof_fetchcfg.log 2>&1; echo "fetch (configs) rc=$?"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_prof_sq -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 > $O/${TAG}_prof_sq.log 2>&1; echo "sq rc=$?"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof_famwrite -- python3 $R/tools/exp/family_front.py > $O/${TAG}_prof_famwrite.log 2>&1; echo "family write rc=$?"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_prof_famfetch -- python3 $R/tools/exp/family_front.py > $O/${TAG}_prof_famfetch.log 2>&1; echo "family fetch rc=$?"
  python3 $R/tools/pmc_aggregate.py $O/${TAG}_prof_fetch $O/${TAG}_prof_write $O/${TAG}_prof_writecfg $O/${TAG}_prof_fetchcfg $O/${TAG}_prof_sq $O/${TAG}_prof_famwrite $O/${TAG}_prof_famfetch
  for i in 1 2 3 4 5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_adi$i -- python3 $R/tools/microbench.py adi expand > $O/${TAG}_prof_adi$i.log 2>&1; echo "adi$i rc=$?"
  done
  if [ -x $R/tools/exp/exp_step2 ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_design -- $R/tools/exp/exp_step2 22 0 1 > $O/${TAG}_design.log 2>&1; echo "design rc=$?"
  fi
  find $O -name "*kernel_trace.csv" -path "*${TAG}_prof*" -delete
  du -sh $O
  cd $R
  python tools/bench_cfg5.py > $O/${TAG}_cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
  python tools/bench_rollout.py > $O/${TAG}_rollout.json 2>/dev/null; echo "rollout rc=$?"
  python tools/bench_facade.py > $O/${TAG}_facade.json 2>/dev/null; echo "facade rc=$?"
  python tools/bench_adi_pipeline.py > $O/${TAG}_adi_pipeline.json 2>/dev/null; echo "adi pipeline rc=$?"
  python tools/bench_legacy_rng.py > $O/${TAG}_legacy_rng.json 2>/dev/null; echo "legacy rc=$?"
fi
