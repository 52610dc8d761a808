#!/bin/bash
# r04: two-level factorisation -- bulk update gated behind the next chain's residency: A/B + C3 bench + C3-size parity
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in 0 1; do
  echo "GPMPC_GATE_BULK=$v"
  GPMPC_GATE_BULK=$v timeout 300 python tools/fit_batch_sweep.py 4096 2>&1 | tail -7
  GPMPC_GATE_BULK=$v timeout 300 python bench.py --config C3 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('C3 ms/step %.1f' % j['ms_per_step'], 'factor ms', j['roofline']['avg_launch_ms'], 'frac', j['roofline']['frac'])"
done
timeout 900 python -m pytest tests -m gpu -x -q -k "c3 or c5 or car_model or synthetic" 2>&1 | tail -4
