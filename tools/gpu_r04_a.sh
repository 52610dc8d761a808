#!/bin/bash
# r04: prediction next to the fit's tail -- parity, same-box A/B of the switches, timeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "behind_tail or c2_full or synthetic or two_handles or timeout" 2>&1 | tail -8
run() {
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  value %.0f  ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f crosscov %.3f solve %.3f frac %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm'], p['crosscov'], p.get('solve',0), j['roofline']['frac']))"
}
for rep in 1 2; do
  GPMPC_EARLY_STATUS=0 GPMPC_ALPHA_SIDE=0 GPMPC_PREDICT_OVERLAP=0 run "baseline          "
  GPMPC_ALPHA_SIDE=0 GPMPC_PREDICT_OVERLAP=0 run "early status      "
  GPMPC_PREDICT_OVERLAP=0 run "early+alpha side  "
  GPMPC_CROSSCOV_WGS=0 run "overlap, cc full  "
  GPMPC_CROSSCOV_WGS=512 run "overlap, cc 512   "
  run "overlap, cc 256   "
  GPMPC_CROSSCOV_WGS=128 run "overlap, cc 128   "
done
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_tl" -o t -- python "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-secondary > "$R/gpurun_out/prof_tl.log" 2>&1; echo "rocprof rc=$?"
python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_tl/t_results.db" > "$R/gpurun_out/r04_step_timeline_b.txt"; tail -32 "$R/gpurun_out/r04_step_timeline_b.txt"
