#!/bin/bash
# Build the emulator flavour of the C-ABI library (TEST INFRASTRUCTURE, see hip/hip_runtime.h):
# the unmodified gp_mpc_amd/csrc sources compiled by g++ against the fiber-based HIP shim.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/_build"
g++ -O2 -std=c++17 -fPIC -shared -Wno-psabi -Wno-unknown-pragmas -I"$HERE" \
    -x c++ "$ROOT/gp_mpc_amd/csrc/gpmpc_api.hip" "$HERE/emu_runtime.cpp" \
    -o "$HERE/_build/libgpmpc_emu.so"
echo "built $HERE/_build/libgpmpc_emu.so"
