#!/bin/bash
# r04 A/B of two builds on one box: gp_mpc_amd/csrc/libgpmpc_hip_prev.so (the previous build, copied there by hand) against the current one;
# hand-off tests, C2 bench pairs, chain + courier stamps of the current build
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
D=gp_mpc_amd/csrc; cp $D/libgpmpc_hip.so /tmp/new.so; cp $D/libgpmpc_hip_prev.so /tmp/old.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cholesky or c2_full or two_handles or random_shapes or without_courier" 2>&1 | tail -3 | sed "s/^/new tests: /"
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  C2 ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f' % (j['ms_per_step'], p['factor'], p['chain'], p['vargemm']))"
}
for rep in 1 2 3; do
  cp /tmp/old.so $D/libgpmpc_hip.so; run "previous    "
  cp /tmp/new.so $D/libgpmpc_hip.so; run "current     "
done
GPMPC_CHAIN_TRACE=gpurun_out/ct.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/chain_trace.py gpurun_out/ct.bin 64 > gpurun_out/r04_chain_trace_poll.txt 2>&1
python tools/worker_trace.py gpurun_out/ct.bin 64 2>&1 | grep -v "^ *[0-9]*a .*-7[0-9][0-9][0-9][0-9][0-9][0-9]" > gpurun_out/r04_worker_trace_poll.txt; rm -f gpurun_out/ct.bin
head -13 gpurun_out/r04_chain_trace_poll.txt; head -8 gpurun_out/r04_worker_trace_poll.txt
