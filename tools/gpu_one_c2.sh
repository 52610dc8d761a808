#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests -m gpu -q -k "c2_full" 2>&1 | grep -E "^E |assert|passed|failed|Error" | head -20
