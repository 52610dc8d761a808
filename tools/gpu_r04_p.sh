#!/bin/bash
# r04: the mean from the persistent variance product's fused reduction (no mean_dot next to it) -- parity, same-box A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "variance_persistent or behind_tail or c2_full or gp_class or two_handles or synthetic or random_shapes or mean_functions" 2>&1 | tail -4
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  value %.0f  ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f frac %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm'], j['roofline']['frac']))"
}
for rep in 1 2 3; do
  GPMPC_FUSED_MEAN=0 run "mean_dot next to the product"
  GPMPC_FUSED_MEAN=1 run "mean fused into the product "
done
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_p" -o t -- python "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > "$R/gpurun_out/prof_p.log" 2>&1
python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_p/t_results.db" > "$R/gpurun_out/r04_step_timeline_fused_mean.txt"; tail -9 "$R/gpurun_out/r04_step_timeline_fused_mean.txt"; rm -rf "$R/gpurun_out/prof_p"
