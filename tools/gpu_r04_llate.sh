#!/bin/bash
# r04: chain kernel, L_kk stored behind the leaf's flag (GPMPC_CHAIN_WT=2: only inv_kk is drained in front of it) vs =1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
GPMPC_CHAIN_WT=2 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cholesky or c2_full or two_handles or random_shapes" 2>&1 | tail -2 | sed "s/^/CHAIN_WT=2 tests: /"
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  C2 ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f' % (j['ms_per_step'], p['factor'], p['chain'], p['vargemm']))"
}
for rep in 1 2 3 4; do
  GPMPC_CHAIN_WT=1 run "CHAIN_WT=1"
  GPMPC_CHAIN_WT=2 run "CHAIN_WT=2"
done
GPMPC_CHAIN_WT=2 GPMPC_CHAIN_TRACE=gpurun_out/ct.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/chain_trace.py gpurun_out/ct.bin 64 2>&1 | head -13; rm -f gpurun_out/ct.bin
