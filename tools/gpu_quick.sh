#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
GPMPC_VERBOSE=1 timeout 900 python -m pytest tests -m gpu -x -q -k "odd_size" -s 2>&1 | grep -E "factor Np|passed|failed|Error|assert" | sort | uniq -c | head
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
