#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for ts in 2 3 4; do
GPMPC_TS=$ts timeout 300 python tools/bench_c3.py 2>/dev/null | grep -E "C5 phases|one pass" | cut -c1-220
done
