#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/bench_c3.py 2>/dev/null | grep -E "rollout|C5 IPOPT" | cut -c1-200
