#!/bin/bash
# Round-5 session A: full GPU test suite, the ADI pipeline split (eager + graph), config 5 / rollout / legacy-RNG figures.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-r05a}
mkdir -p $O
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/${TAG}_pytest.log
  tail -5 $O/${TAG}_pytest.log
fi
bash $R/tools/gpu_adi_split.sh ${TAG} || true
bash $R/tools/gpu_adi_split.sh ${TAG}g --graph || true
cd $R
python tools/bench_cfg5.py > $O/${TAG}_cfg5.json 2> $O/${TAG}_cfg5.err; echo "cfg5 rc=$?"; cat $O/${TAG}_cfg5.json
python tools/bench_rollout.py > $O/${TAG}_rollout.json 2>/dev/null; echo "rollout rc=$?"
python tools/bench_legacy_rng.py > $O/${TAG}_legacy.json 2> $O/${TAG}_legacy.err; echo "legacy rc=$?"; cat $O/${TAG}_legacy.json
