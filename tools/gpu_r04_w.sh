#!/bin/bash
# r04: w = L^-1 y next to the inverse's last product (GPMPC_EARLY_W=1, factor_chain) vs behind it on the workers' queue (=0):
# behind-the-tail parity tests, same-box C2 A/B, step timeline of the new default
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tail or c2_full or synthetic or two_handles or timeout or mean_func or strict" 2>&1 | tail -3 | sed "s/^/EARLY_W=1 tests: /"
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  C2 ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f solve %.3f' % (j['ms_per_step'], p['factor'], p['chain'], p['vargemm'], p['solve']))"
}
for rep in 1 2 3; do
  GPMPC_EARLY_W=0 run "EARLY_W=0"
  GPMPC_EARLY_W=1 run "EARLY_W=1"
done
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_w" -o t -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > "$R/gpurun_out/prof_w.log" 2>&1; echo "rocprof rc=$?"
python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_w/t_results.db" > "$R/gpurun_out/r04_step_timeline_early_w.txt" 2>&1; rm -rf "$R/gpurun_out/prof_w"
sed -n 28,60p "$R/gpurun_out/r04_step_timeline_early_w.txt"
