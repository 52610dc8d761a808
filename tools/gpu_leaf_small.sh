#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for N in 512 1024 2048; do
GPMPC_CHAIN_TRACE=gpurun_out/ct.bin timeout 120 python bench.py --N $N --B 1000 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
echo "== N=$N"; python tools/chain_trace.py gpurun_out/ct.bin $((N/64)) | tail -17
done
rm -f gpurun_out/ct.bin
