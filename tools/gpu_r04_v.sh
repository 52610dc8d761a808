#!/bin/bash
# r04: chain kernel, deferred publications (GPMPC_CHAIN_WT=2: the write-through stores of L_kk / inv_kk drain behind the panel
# row's products, those of L(k+1,k) behind the diagonal block's update, when the next tiles have landed already) vs =1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
GPMPC_CHAIN_WT=2 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cholesky or c2_full or synthetic or two_handles or timeout or boundary or random_shapes" 2>&1 | tail -3 | sed "s/^/CHAIN_WT=2 tests: /"
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  C2 ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f' % (j['ms_per_step'], p['factor'], p['chain'], p['vargemm']))"
}
for rep in 1 2 3; do
  GPMPC_CHAIN_WT=1 run "CHAIN_WT=1"
  GPMPC_CHAIN_WT=2 run "CHAIN_WT=2"
done
for wt in 2; do
  GPMPC_CHAIN_WT=$wt GPMPC_CHAIN_TRACE=gpurun_out/ct.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  python tools/chain_trace.py gpurun_out/ct.bin 64 > gpurun_out/r04_chain_trace_cwt$wt.txt 2>&1
  python tools/worker_trace.py gpurun_out/ct.bin 64 2>&1 | grep -v "^ *[0-9]*a .*-7[0-9][0-9][0-9][0-9][0-9][0-9]" > gpurun_out/r04_worker_trace_cwt$wt.txt; rm -f gpurun_out/ct.bin
  echo "--- CHAIN_WT=$wt"; head -13 gpurun_out/r04_chain_trace_cwt$wt.txt
done
