#!/bin/bash
# r04: lock-step restart search -- C4 tests + bench (lock-step vs one restart after the other)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "c4 or training" --durations=8 2>&1 | tail -8
GPMPC_VERBOSE=1 timeout 300 python bench.py --config C4 --steps 1 --warmup 0 2>&1 | grep "lock-step batch" | sed 's/gpmpc: lock-step batch of //' | tr '\n' ';' | cut -c1-1800
echo
for v in 1 0; do
  GPMPC_TRAIN_LOCKSTEP=$v timeout 300 python bench.py --config C4 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('LOCKSTEP=$v restarts/s %.1f  ms/step %.1f best %.6f finite %d evals %d' % (j['value'], j['ms_per_step'], j['best_nll'], j['finite_restarts'], j['evaluations_this_rank']))"
done
