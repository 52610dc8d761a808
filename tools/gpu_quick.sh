#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/bench_append.py 2>&1 | grep -v amdgpu.ids
