#!/bin/bash
# r04: panel inverses that follow the chain block by block (flag-gated merge launches) -- parity, A/B of the panel mask and of
# the fourth queue's priority, timelines
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "cholesky or c2_full or synthetic or behind_tail or two_handles or timeout or boundary or random_shapes or append" 2>&1 | tail -5
run() {
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  value %.0f  ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f crosscov %.3f solve %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm'], p['crosscov'], p.get('solve',0)))"
}
for rep in 1 2; do
  GPMPC_TRTRI_INCR=0 run "classic              "
  GPMPC_TRTRI_INCR=4 run "last panel, low prio "
  GPMPC_TRTRI_INCR=6 run "panels 1 2, low prio "
  GPMPC_TRTRI_INCR=6 GPMPC_BULK_PRIORITY=0 run "panels 1 2, normal   "
  GPMPC_TRTRI_INCR=7 run "all panels, low prio "
  GPMPC_TRTRI_INCR=7 GPMPC_BULK_PRIORITY=0 run "all panels, normal   "
  GPMPC_TRTRI_INCR=0 GPMPC_BULK_PRIORITY=0 run "classic, normal      "
done
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
for cfg in "6 1" "6 0" "7 0"; do
  set -- $cfg
  GPMPC_TRTRI_INCR=$1 GPMPC_BULK_PRIORITY=$2 timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_f_$1_$2" -o t -- python "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > "$R/gpurun_out/prof_f_$1_$2.log" 2>&1; echo "rocprof incr=$1 prio=$2 rc=$?"
  python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_f_$1_$2/t_results.db" > "$R/gpurun_out/r04_step_timeline_incr$1_prio$2.txt"
  grep -n "chol_chain_kernel\|vargemm_persist\|crosscov" "$R/gpurun_out/r04_step_timeline_incr$1_prio$2.txt" | head -4
  rm -rf "$R/gpurun_out/prof_f_$1_$2"
done
