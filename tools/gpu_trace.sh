#!/bin/bash
# in-kernel time stamps of the chain + workers + courier for one bench configuration
mkdir -p gpurun_out
GPMPC_CHAIN_TRACE=gpurun_out/chain_trace.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python tools/chain_trace.py gpurun_out/chain_trace.bin 64 2>&1 | tee gpurun_out/chain_trace.txt
python tools/worker_trace.py gpurun_out/chain_trace.bin 64 2>&1 | tee gpurun_out/worker_trace.txt | head -50
python tools/courier_trace.py gpurun_out/chain_trace.bin 64 2>&1 | tee gpurun_out/courier_trace.txt
rm -f gpurun_out/chain_trace.bin
