#!/bin/bash
# kernel trace of a few bench steps; tools/step_timeline.py prints one step as a timeline per queue
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_tl" -o t -- python "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline > "$R/gpurun_out/prof_tl.log" 2>&1; echo "rocprof rc=$?"
python "$R/tools/step_timeline.py" "$R/gpurun_out/prof_tl/t_results.db"
