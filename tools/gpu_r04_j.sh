#!/bin/bash
# r04: tunables of the two-level factorisation at the restart search's shape (batches of up to 64 matrices of 4096^2)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
c4() {
  timeout 300 python bench.py --config C4 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 restarts/s %.1f  ms/step %.1f best %.9f evals %d' % (j['value'], j['ms_per_step'], j['best_nll'], j['evaluations_this_rank']))"
}
c4 "default (W = 8)        "
GPMPC_TWOLEVEL=4 c4 "W = 4                  "
GPMPC_TWOLEVEL=16 c4 "W = 16                 "
GPMPC_LOOKAHEAD=0 c4 "no look-ahead          "
GPMPC_GATE_BULK=0 c4 "bulk not gated         "
GPMPC_T64_STAGES=3 c4 "64-row tiles: 3 images "
GPMPC_T64_STAGES=2 c4 "64-row tiles: 2 images "
GPMPC_T128=256 c4 "128-row tiles from 256 "
GPMPC_T128=2048 c4 "128-row tiles from 2048"
c4 "default (W = 8) again  "
