#!/bin/bash
# tools/build_ab_lib.sh <commit> <name>: build libgpmpc_hip.so of another commit into tools/ab_libs/<name>.so (for tools/gpu_ab.py lib=<name>)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
T=$(mktemp -d)
git -C "$ROOT" archive "$1" gp_mpc_amd/csrc include | tar -x -C "$T"
make -C "$T/gp_mpc_amd/csrc" libgpmpc_hip.so > /dev/null 2>&1
mkdir -p "$ROOT/tools/ab_libs"
cp "$T/gp_mpc_amd/csrc/libgpmpc_hip.so" "$ROOT/tools/ab_libs/$2.so"
rm -rf "$T"
echo "built tools/ab_libs/$2.so from $1"
