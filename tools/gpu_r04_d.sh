#!/bin/bash
# r04: persistent variance product (static tile schedule) -- parity, same-box A/B against the dispatcher's order, timeline, fetched bytes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "variance_persistent or behind_tail or callback_classes or c2_full or bench" 2>&1 | tail -6
run() {
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  value %.0f  ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f crosscov %.3f solve %.3f frac %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm'], p['crosscov'], p.get('solve',0), j['roofline']['frac']))"
}
for rep in 1 2 3; do
  GPMPC_VARGEMM_PERSIST=0 run "dispatcher order "
  GPMPC_VARGEMM_PERSIST=1 run "static schedule  "
done
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
GPMPC_VERBOSE=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_vp" -o t -- python "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > "$R/gpurun_out/prof_vp.log" 2>&1; echo "rocprof rc=$?"
grep "variance schedule" "$R/gpurun_out/prof_vp.log" | head -2
python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_vp/t_results.db" > "$R/gpurun_out/r04_step_timeline_vp.txt"; tail -9 "$R/gpurun_out/r04_step_timeline_vp.txt"
for C in FETCH_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$R/gpurun_out/pmc_vp_$C" -o p -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$R/gpurun_out/pmc_vp_$C.log" 2>&1
  echo "== $C rc=$?"
  python "$R/tools/pmc_summary.py" "$R/gpurun_out/pmc_vp_$C/p_results.db" > "$R/gpurun_out/r04_pmc_vp_$C.txt" 2>&1; grep -i "vargemm\|128, 128" "$R/gpurun_out/r04_pmc_vp_$C.txt" | head -4
done
