#!/bin/bash
# a subset of the GPU tier: tools/gpu_one_test.sh "<-k expression>"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$1" 2>&1 | tail -4
