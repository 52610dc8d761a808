#!/bin/bash
# r04: cross-covariances of the prediction behind a fit's tail as write-through stores (GPMPC_CROSSCOV_WT=1) vs plain (=0):
# the tail's small kernels pay at their end for the write-back of whatever sits dirty in the L2s
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tail or c2_full" 2>&1 | tail -2 | sed "s/^/CROSSCOV_WT=1 tests: /"
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  C2 ms/step %.3f  factor %.3f  chain %.3f crosscov %.3f vargemm %.3f' % (j['ms_per_step'], p['factor'], p['chain'], p['crosscov'], p['vargemm']))"
}
for rep in 1 2 3; do
  GPMPC_CROSSCOV_WT=0 run "CROSSCOV_WT=0"
  GPMPC_CROSSCOV_WT=1 run "CROSSCOV_WT=1"
done
