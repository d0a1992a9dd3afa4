#!/bin/bash
# One GPU session: tests, bench, rocprof stats + PMC passes, fresh-process ADI repeats, design A/B.
# Run via gpurun from the repo root:   gpurun --timeout 1190 -- 'SKIP_TESTS=1 bash tools/gpu_round.sh r03a'
# rocprofv3 always gets the program itself after `--` (python3 / a binary), never a wrapper.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r04}
mkdir -p $O
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/${TAG}_pytest.log
  tail -3 $O/${TAG}_pytest.log
fi
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
# (1) kernel trace + stats of the bench command: the headline alone (its k_step average is the one roofline.launch_us must agree
# with; the configs launch the same instantiation with a reward array), then with every config's kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_stats -- python3 $R/bench.py --no-cpu --no-configs --steps 200 --warmup 20 > $O/${TAG}_prof_stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_statscfg -- python3 $R/bench.py --no-cpu --steps 200 --warmup 20 > $O/${TAG}_prof_statscfg.log 2>&1; echo "stats (configs) rc=$?"
# (2) PMC passes, one counter group per run (FETCH_SIZE and WRITE_SIZE do not fit one pass).  The traffic figure of the headline
# kernel comes from runs WITHOUT the extra configs: they launch the same k_step instantiation with a reward array (+4 B per
# cube), which would pollute the per-kernel mean.  A second WRITE_SIZE pass with the configs covers k_adi / k_expand / dense.
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_prof_fetch -- python3 $R/bench.py --no-cpu --no-configs --steps 10 --warmup 2 > $O/${TAG}_prof_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof_write -- python3 $R/bench.py --no-cpu --no-configs --steps 10 --warmup 2 > $O/${TAG}_prof_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${TAG}_prof_writecfg -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 > $O/${TAG}_prof_writecfg.log 2>&1; echo "write (configs) rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${TAG}_prof_fetchcfg -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 > $O/${TAG}_prof_fetchcfg.log 2>&1; echo "fetch (configs) rc=$?"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O/${TAG}_prof_sq -- python3 $R/bench.py --no-cpu --steps 10 --warmup 2 > $O/${TAG}_prof_sq.log 2>&1; echo "sq rc=$?"
# (3) the ADI kernel in FIVE fresh processes (placement-to-placement spread was 15 % in round 1)
for i in 1 2 3 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_adi$i -- python3 $R/tools/microbench.py adi expand > $O/${TAG}_prof_adi$i.log 2>&1; echo "adi$i rc=$?"
done
# (4) design A/B harness: literal north_star design vs the shipped select network vs copies (kernel-stat rows)
if [ -x $R/tools/exp/exp_step2 ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_design -- $R/tools/exp/exp_step2 22 0 1 > $O/${TAG}_design.log 2>&1; echo "design rc=$?"
fi
cd $R
python tools/bench_cfg5.py > $O/${TAG}_cfg5.json 2>/dev/null; echo "cfg5 rc=$?"
python tools/bench_rollout.py > $O/${TAG}_rollout.json 2>/dev/null; echo "rollout rc=$?"
python tools/bench_facade.py > $O/${TAG}_facade.json 2>/dev/null; echo "facade rc=$?"
python tools/bench_adi_pipeline.py > $O/${TAG}_adi_pipeline.json 2>/dev/null; echo "adi pipeline rc=$?"
find $O -name "*.csv" | grep ${TAG} | wc -l
