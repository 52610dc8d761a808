#!/bin/bash
# run the GPU tests matching $1 (pytest -k), full failure text to gpurun_out/one.log
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
NCCL_DEBUG=${NCCL_DEBUG:-} timeout 900 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tee gpurun_out/one.log | grep -E "Error|error|passed|failed|NCCL WARN" | head -40
