#!/bin/bash
# r04: the courier's three tiles as 16-byte write-through stores (lane-pair exchange, GPMPC_WORKER_WT=2) vs 8-byte ones (=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
GPMPC_WORKER_WT=2 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cholesky or c2_full or synthetic or two_handles or boundary or random_shapes" 2>&1 | tail -3 | sed "s/^/WORKER_WT=2 tests: /"
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  C2 ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f' % (j['ms_per_step'], p['factor'], p['chain'], p['vargemm']))"
}
for rep in 1 2 3; do
  GPMPC_WORKER_WT=1 run "WORKER_WT=1"
  GPMPC_WORKER_WT=2 run "WORKER_WT=2"
done
GPMPC_WORKER_WT=2 GPMPC_CHAIN_TRACE=gpurun_out/ct.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/chain_trace.py gpurun_out/ct.bin 64 2>&1 | head -13
python tools/courier_trace.py gpurun_out/ct.bin 64 2>&1 | tail -17; rm -f gpurun_out/ct.bin
