#!/bin/bash
# r04: cross-covariances of the prediction behind the tail in two parts (a throttled one next to the level launches, the rest behind them)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() {
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  value %.0f  ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f crosscov %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm'], p['crosscov']))"
}
for rep in 1 2; do
  run "one launch at chain end (256 wgs)"
  GPMPC_CROSSCOV_SPLIT_PCT=30 run "30pct on 64 wgs, rest after levels "
  GPMPC_CROSSCOV_SPLIT_PCT=50 run "50pct on 64 wgs, rest after levels "
  GPMPC_CROSSCOV_SPLIT_PCT=50 GPMPC_CROSSCOV_A_WGS=128 run "50pct on 128 wgs, rest after level"
  GPMPC_CROSSCOV_SPLIT_PCT=70 GPMPC_CROSSCOV_A_WGS=128 run "70pct on 128 wgs, rest after level"
done
