#!/bin/bash
# the C2 bench line + the kernel trace, step timeline and chain stamps of the same command on the same box (a subset of gpu_refresh_r04.sh)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pair
timeout 600 python bench.py > gpurun_out/pair/r04_bench.json 2> gpurun_out/pair/r04_bench.err; python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/pair/r04_bench.json') if l.startswith('{')][-1])
print('C2 value %.0f ms/step %.3f (all brackets %.3f) frac %.3f ubench %.1f phases %s' % (j['value'], j['ms_per_step'], j['ms_per_step_all_brackets'], j['roofline']['frac'], j['roofline']['peak_measured_mfma_only_ubench'], {k: round(v,3) for k,v in j['phases_ms_per_step'].items()}))
print('cpu', j.get('cpu_baseline',{}).get('value'), 'secondary c3 ms', j['secondary']['c3']['ms_per_step'], 'c4 restarts/s', j['secondary']['c4']['restarts_per_s'])
PY
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_pair" -o t -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > "$R/gpurun_out/prof_pair.log" 2>&1; echo "rocprof rc=$?"
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_pair/t_results.db" --steps 6 > "$R/gpurun_out/pair/r04_kernel_trace_bench.txt"; head -6 "$R/gpurun_out/pair/r04_kernel_trace_bench.txt"
python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_pair/t_results.db" > "$R/gpurun_out/pair/r04_step_timeline.txt" 2>&1; rm -rf "$R/gpurun_out/prof_pair"
cd "$R"
GPMPC_CHAIN_TRACE=gpurun_out/ct.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python tools/chain_trace.py gpurun_out/ct.bin 64 > gpurun_out/pair/r04_chain_trace.txt 2>&1; head -13 gpurun_out/pair/r04_chain_trace.txt | tail -5
python tools/worker_trace.py gpurun_out/ct.bin 64 2>&1 | grep -v "^ *[0-9]*a .*-7[0-9][0-9][0-9][0-9][0-9][0-9]" > gpurun_out/pair/r04_worker_trace.txt; rm -f gpurun_out/ct.bin
