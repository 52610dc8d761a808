#!/bin/bash
# r04: one courier per stage (GPMPC_COURIER=2: L(k+2,k) / the two hand-off tiles) vs the one courier (=1)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_couriers or cholesky or c2_full" 2>&1 | tail -3 | sed "s/^/tests: /"
GPMPC_COURIER=2 GPMPC_VERBOSE=1 timeout 120 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary 2>&1 | grep -m2 "gpmpc: factor"
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  C2 ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f' % (j['ms_per_step'], p['factor'], p['chain'], p['vargemm']))"
}
for rep in 1 2 3; do
  GPMPC_COURIER=1 run "COURIER=1"
  GPMPC_COURIER=2 run "COURIER=2"
done
GPMPC_COURIER=2 GPMPC_CHAIN_TRACE=gpurun_out/ct.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/chain_trace.py gpurun_out/ct.bin 64 2>&1 | head -13
python tools/courier_trace.py gpurun_out/ct.bin 64 2>&1 | tail -17; rm -f gpurun_out/ct.bin
