#!/bin/bash
# r04: workgroups of the throttled cross-covariance launch next to the tail (GPMPC_CROSSCOV_WGS; default = CU count), after the tail got shorter
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  C2 ms/step %.3f  factor %.3f  chain %.3f crosscov %.3f vargemm %.3f' % (j['ms_per_step'], p['factor'], p['chain'], p['crosscov'], p['vargemm']))"
}
for rep in 1 2; do
  run "default (256)   "
  for n in 384 512 1024; do GPMPC_CROSSCOV_WGS=$n run "CROSSCOV_WGS=$n"; done
done
