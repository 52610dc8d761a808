#!/bin/bash
# r04: cross-covariances of the prediction behind the tail started after the last panel's level launches instead of at chain end
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() {
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  value %.0f  ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f crosscov %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm'], p['crosscov']))"
}
for rep in 1 2; do
  run "at chain end, 256 wgs       "
  GPMPC_CROSSCOV_AFTER_LEVELS=1 run "after levels, 256 wgs       "
  GPMPC_CROSSCOV_AFTER_LEVELS=1 GPMPC_CROSSCOV_WGS=512 run "after levels, 512 wgs       "
  GPMPC_CROSSCOV_AFTER_LEVELS=1 GPMPC_CROSSCOV_WGS=0 run "after levels, unthrottled   "
done
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
GPMPC_CROSSCOV_AFTER_LEVELS=1 GPMPC_CROSSCOV_WGS=0 timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_m" -o t -- python "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > "$R/gpurun_out/prof_m.log" 2>&1
python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_m/t_results.db" > "$R/gpurun_out/r04_step_timeline_cc_after_levels.txt"; tail -22 "$R/gpurun_out/r04_step_timeline_cc_after_levels.txt"; rm -rf "$R/gpurun_out/prof_m"
