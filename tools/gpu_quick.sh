#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
C3_N=4096 GPMPC_VERBOSE=1 timeout 300 python tools/bench_c3.py 2>&1 | grep -E "C3 fit|factor Np" | sort | uniq -c | cut -c1-330
C3_N=2048 GPMPC_VERBOSE=1 timeout 300 python tools/bench_c3.py 2>&1 | grep -E "C3 fit|factor Np" | sort | uniq -c | cut -c1-330
