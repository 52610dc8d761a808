#!/bin/bash
# kernel timeline of one C2 step under rocprofv3 (tools/step_timeline.py)
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$R/gpurun_out/prof_tl" -o t -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > "$R/gpurun_out/prof_tl.log" 2>&1; echo "rocprof rc=$?"
python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_tl/t_results.db" > "$R/gpurun_out/step_timeline.txt" 2>&1; tail -14 "$R/gpurun_out/step_timeline.txt"
