#!/bin/bash
# r04 baseline at the start of the second session: GPU tier, smoke, bench line (C2 + secondary), kernel trace + step timeline,
# chain / worker stamps, kernel trace of one C4 step (where the lock-step search spends its time).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r04_gpu_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r04_gpu_tests.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; tail -c 1500 gpurun_out/r04_bench.json; echo
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r04" -o t -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > "$R/gpurun_out/prof_r04.log" 2>&1; echo "rocprof rc=$?"
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_r04/t_results.db" --steps 6 > "$R/gpurun_out/r04_kernel_trace_bench.txt"; head -14 "$R/gpurun_out/r04_kernel_trace_bench.txt"
python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_r04/t_results.db" > "$R/gpurun_out/r04_step_timeline.txt" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r04c4" -o t -- python "$R/bench.py" --config C4 --steps 1 --warmup 1 > "$R/gpurun_out/prof_r04c4.log" 2>&1; echo "rocprof C4 rc=$?"
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_r04c4/t_results.db" --steps 2 > "$R/gpurun_out/r04_kernel_trace_bench_c4.txt"; head -24 "$R/gpurun_out/r04_kernel_trace_bench_c4.txt"
cd "$R"
timeout 600 python bench.py --config C4 > gpurun_out/r04_bench_c4.json 2>gpurun_out/c4.err; cut -c1-500 gpurun_out/r04_bench_c4.json
GPMPC_CHAIN_TRACE=gpurun_out/chain_trace_r04.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/chain_trace.py gpurun_out/chain_trace_r04.bin 64 > gpurun_out/r04_chain_trace.txt 2>&1; tail -14 gpurun_out/r04_chain_trace.txt
python tools/worker_trace.py gpurun_out/chain_trace_r04.bin 64 2>&1 | grep -v "^ *[0-9]*a .*-7[0-9][0-9][0-9][0-9][0-9][0-9]" > gpurun_out/r04_worker_trace.txt; rm -f gpurun_out/chain_trace_r04.bin
rm -rf gpurun_out/prof_r04/*.db.bak 2>/dev/null; du -sh gpurun_out | tail -1
