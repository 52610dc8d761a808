#!/bin/bash
# r04: K^-1 = X^T X on a transposed copy through the persistent static-schedule kernel; lock-step search with value-only trials and
# gradients from the retained factors -- parity (C4 / training / K^-1 users), C4 and C3 A/B on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "c4 or lockstep or training or train or variance_persistent or c3 or car_model or moment or em_sens or nll or gp_class or tank" --durations=5 2>&1 | tail -12
c4() {
  timeout 300 python bench.py --config C4 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 restarts/s %.1f  ms/step %.1f best %.9f finite %d evals %d' % (j['value'], j['ms_per_step'], j['best_nll'], j['finite_restarts'], j['evaluations_this_rank']))"
}
GPMPC_TRAIN_RETAIN=1 GPMPC_VARGEMM_PERSIST=1 c4 "retain, persistent K^-1      "
GPMPC_TRAIN_RETAIN=0 GPMPC_VARGEMM_PERSIST=1 c4 "all gradients, persistent   "
GPMPC_TRAIN_RETAIN=1 GPMPC_VARGEMM_PERSIST=0 c4 "retain, one tile per wg     "
GPMPC_TRAIN_RETAIN=0 GPMPC_VARGEMM_PERSIST=0 c4 "all gradients, one tile     "
GPMPC_TRAIN_RETAIN=1 GPMPC_VARGEMM_PERSIST=1 c4 "retain, persistent K^-1 (2) "
GPMPC_VERBOSE=1 timeout 300 python bench.py --config C4 --steps 1 --warmup 0 2>&1 | grep "lock-step\|schedule" | sed 's/gpmpc: lock-step //' | head -60 > gpurun_out/r04_c4_batches.txt; head -24 gpurun_out/r04_c4_batches.txt
c3() {
  timeout 300 python bench.py --config C3 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 C3 ms/step %.1f' % j['ms_per_step'], 'phases', {k: round(v,2) for k,v in j.get('phases_ms_per_step',{}).items()})"
}
GPMPC_VARGEMM_PERSIST=1 c3 "persistent"
GPMPC_VARGEMM_PERSIST=0 c3 "one tile  "
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_e_c4" -o t -- python "$R/bench.py" --config C4 --steps 1 --warmup 1 > "$R/gpurun_out/prof_e_c4.log" 2>&1; echo "rocprof C4 rc=$?"
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_e_c4/t_results.db" --steps 2 > "$R/gpurun_out/r04_kernel_trace_bench_c4_b.txt"; head -16 "$R/gpurun_out/r04_kernel_trace_bench_c4_b.txt"
