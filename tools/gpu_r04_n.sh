#!/bin/bash
# r04: the small levels of the panel inverses in resident launches that follow the chain (trtri_follow.hpp) -- parity, A/B of the
# panel mask and the workgroup count, timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "cholesky or c2_full or synthetic or behind_tail or two_handles or timeout or boundary or random_shapes or append" 2>&1 | tail -4
run() {
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  value %.0f  ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f crosscov %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm'], p['crosscov']))"
}
for rep in 1 2; do
  GPMPC_TRTRI_FOLLOW=0 run "classic levels         "
  GPMPC_TRTRI_FOLLOW=4 run "last panel follows     "
  GPMPC_TRTRI_FOLLOW=6 run "panels 1, 2 follow     "
  GPMPC_TRTRI_FOLLOW=7 run "all panels follow      "
  GPMPC_TRTRI_FOLLOW=7 GPMPC_TRTRI_FOLLOW_WGS=8 run "all, 8 workgroups      "
  GPMPC_TRTRI_FOLLOW=7 GPMPC_TRTRI_FOLLOW_WGS=32 run "all, 32 workgroups     "
  GPMPC_TRTRI_FOLLOW=7 GPMPC_CROSSCOV_WGS=0 run "all, crosscov unthrottl"
done
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
GPMPC_TRTRI_FOLLOW=7 timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_n" -o t -- python "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > "$R/gpurun_out/prof_n.log" 2>&1
python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_n/t_results.db" > "$R/gpurun_out/r04_step_timeline_follow.txt"; grep -v "32, 32, 2, 1" "$R/gpurun_out/r04_step_timeline_follow.txt" | tail -30; rm -rf "$R/gpurun_out/prof_n"
