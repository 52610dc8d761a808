#!/bin/bash
# r04: chain kernel publications as write-through (sc1) stores + flag (GPMPC_CHAIN_WT=1) vs release fence (=0):
# factorisation parity tests under both, same-box C2/C3 A/B, chain trace of both
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for wt in 1 0; do
  GPMPC_CHAIN_WT=$wt timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cholesky or c2_full or synthetic or two_handles or timeout or boundary or random_shapes or c3" 2>&1 | tail -3 | sed "s/^/WT=$wt tests: /"
done
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  C2 ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f' % (j['ms_per_step'], p['factor'], p['chain'], p['vargemm']))"
}
c3() { timeout 300 python bench.py --config C3 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 C3 ms/step %.1f' % j['ms_per_step'], {k: round(v,2) for k,v in j.get('phases_ms_per_step',{}).items()})"; }
for rep in 1 2 3; do
  GPMPC_CHAIN_WT=0 run "WT=0"
  GPMPC_CHAIN_WT=1 run "WT=1"
done
GPMPC_CHAIN_WT=0 c3 "WT=0"; GPMPC_CHAIN_WT=1 c3 "WT=1"
for wt in 0 1; do
  GPMPC_CHAIN_WT=$wt timeout 300 python tools/chain_trace.py > gpurun_out/r04_chain_trace_wt$wt.txt 2>&1
  echo "--- chain trace WT=$wt"; tail -25 gpurun_out/r04_chain_trace_wt$wt.txt
done
