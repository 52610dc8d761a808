#!/bin/bash
# Runs one micro-benchmark of tools/ubench on the GPU box: tools/gpu_ubench.sh <name> [args...]   (e.g. `gpurun -- tools/gpu_ubench.sh mfma_issue_bench`)
# The binaries are built here first (hipcc --offload-arch=gfx950 -O2 tools/ubench/<name>.hip -o tools/ubench/<name>); they travel with gpurun.
cd "$GRAFT_REPO_ROOT"; N=$1; shift
[ -x tools/ubench/$N ] || hipcc --offload-arch=gfx950 -O2 tools/ubench/$N.hip -o tools/ubench/$N || exit 1
timeout 600 tools/ubench/$N "$@"
