#!/bin/bash
# chain kernel A/B: parity tests of the factorisation paths, bench line (3 runs), in-kernel time stamps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "chol or c2_full or worker or tank or car or jitter or golden or timeout or two_handles or c3_size or random_shapes" 2>&1 | grep -E "passed|failed|rror" | tail -3
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('value %.0f  ms/step %.3f  factor %.3f  chain %.3f  vargemm %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm']))"; done
GPMPC_CHAIN_TRACE=gpurun_out/chain_trace_ab.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python tools/chain_trace.py gpurun_out/chain_trace_ab.bin 64 2>&1 | tail -13
