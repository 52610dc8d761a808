#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for pr in 0 1 3; do timeout 300 tools/ubench/vargemm_bench_p$pr; done
