#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for ab in 0 1 3 4 7 8 12 15; do timeout 120 tools/ubench/gemm_ablate_$ab; done
