#!/bin/bash
# round-4 refresh of the judged artefacts: GPU tier, smoke, bench lines (C2 with CPU baseline and secondary, C3, C4), kernel traces
# of the C2 and C3 steps + the C2 step timeline, chain / worker stamps, C5.  Everything lands in gpurun_out/r04_*.
# (PMC passes of the dominant kernel: tools/gpu_r04_h.sh -> profiles/r04_pmc_*, r04_traffic.json; of the EM pair sums: gpu_r04_g.sh)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r04_gpu_tests.txt 2>&1; grep -E "passed|failed|error" gpurun_out/r04_gpu_tests.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r04_bench.json') if l.startswith('{')][-1])
print('C2 value %.0f ms/step %.3f (all brackets %.3f) frac %.3f phases %s' % (j['value'], j['ms_per_step'], j['ms_per_step_all_brackets'], j['roofline']['frac'], {k: round(v,3) for k,v in j['phases_ms_per_step'].items()}))
print('cpu', j.get('cpu_baseline',{}).get('value'), 'secondary c3 ms', j['secondary']['c3']['ms_per_step'], 'c4 restarts/s', j['secondary']['c4']['restarts_per_s'])
PY
timeout 600 python bench.py --config C3 > gpurun_out/r04_bench_c3.json 2> gpurun_out/r04_bench_c3.err; cut -c1-300 gpurun_out/r04_bench_c3.json
timeout 600 python bench.py --config C4 > gpurun_out/r04_bench_c4.json 2>gpurun_out/c4.err; cut -c1-300 gpurun_out/r04_bench_c4.json
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r04" -o t -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > "$R/gpurun_out/prof_r04.log" 2>&1; echo "rocprof rc=$?"
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_r04/t_results.db" --steps 6 > "$R/gpurun_out/r04_kernel_trace_bench.txt"; head -12 "$R/gpurun_out/r04_kernel_trace_bench.txt"
python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_r04/t_results.db" > "$R/gpurun_out/r04_step_timeline.txt" 2>&1; rm -rf "$R/gpurun_out/prof_r04"
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r04c3" -o t -- python "$R/bench.py" --config C3 --steps 2 --warmup 1 > "$R/gpurun_out/prof_r04c3.log" 2>&1; echo "rocprof C3 rc=$?"
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_r04c3/t_results.db" --steps 3 > "$R/gpurun_out/r04_kernel_trace_bench_c3.txt"; head -8 "$R/gpurun_out/r04_kernel_trace_bench_c3.txt"; rm -rf "$R/gpurun_out/prof_r04c3"
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r04c4" -o t -- python "$R/bench.py" --config C4 --steps 1 --warmup 1 > "$R/gpurun_out/prof_r04c4.log" 2>&1; echo "rocprof C4 rc=$?"
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_r04c4/t_results.db" --steps 2 > "$R/gpurun_out/r04_kernel_trace_bench_c4.txt"; rm -rf "$R/gpurun_out/prof_r04c4"
cd "$R"
timeout 600 python tools/bench_c3.py 2>gpurun_out/c3.err | grep "C5" > gpurun_out/r04_c5.jsonl; cut -c1-250 gpurun_out/r04_c5.jsonl
GPMPC_CHAIN_TRACE=gpurun_out/chain_trace_r04.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/chain_trace.py gpurun_out/chain_trace_r04.bin 64 > gpurun_out/r04_chain_trace.txt 2>&1; head -13 gpurun_out/r04_chain_trace.txt | tail -5
python tools/worker_trace.py gpurun_out/chain_trace_r04.bin 64 2>&1 | grep -v "^ *[0-9]*a .*-7[0-9][0-9][0-9][0-9][0-9][0-9]" > gpurun_out/r04_worker_trace.txt; rm -f gpurun_out/chain_trace_r04.bin
# soak (write-through hand-offs, joint polls): 1000 consecutive steps, any hand-off time-out prints a message
timeout 300 python bench.py --steps 1000 --warmup 3 --no-cpu-baseline --no-secondary 2>gpurun_out/soak_err.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('soak 1000 steps: value %8.0f  ms/step %.3f  factor %.3f vargemm %.3f' % (d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor'], d['phases_ms_per_step']['vargemm']))" | tee gpurun_out/r04_soak.txt
echo "time-outs: $(grep -c 'timed out' gpurun_out/soak_err.log)" | tee -a gpurun_out/r04_soak.txt
