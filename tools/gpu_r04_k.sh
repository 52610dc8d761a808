#!/bin/bash
# r04: value batches of the lock-step search without L^-1 (formed for accepted points only) -- parity + same-box A/B
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -k "c4 or lockstep or training or train or nll" 2>&1 | tail -4
c4() {
  timeout 300 python bench.py --config C4 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 restarts/s %.1f  ms/step %.1f best %.9f evals %d' % (j['value'], j['ms_per_step'], j['best_nll'], j['evaluations_this_rank']))"
}
for rep in 1 2; do
  GPMPC_TRAIN_SKIP_INVERSE=0 c4 "value batches with L^-1   "
  GPMPC_TRAIN_SKIP_INVERSE=1 c4 "value batches without L^-1"
done
GPMPC_VERBOSE=1 timeout 300 python bench.py --config C4 --steps 1 --warmup 0 2>&1 | grep "lock-step" | sed 's/gpmpc: lock-step //' | head -70 > gpurun_out/r04_c4_batches_b.txt; head -12 gpurun_out/r04_c4_batches_b.txt
