#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 300 tools/ubench/mfma_issue_bench
