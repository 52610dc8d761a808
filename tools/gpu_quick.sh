#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python tools/bench_c3.py 2>/dev/null > gpurun_out/c3.jsonl; cut -c1-120 gpurun_out/c3.jsonl
