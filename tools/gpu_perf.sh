#!/bin/bash
# perf iteration: bench + kernel-trace profile summary.  usage: tools/gpu_perf.sh <tag>
TAG=${1:-x}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"
python - <<PY
import json
l=[x for x in open('gpurun_out/bench_$TAG.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('value %.0f pred/s  ms/step %.3f  roofline %.1f TF  phases %s' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], {k: round(v,3) for k,v in d['phases_ms_per_step'].items()}))
else:
    print(open('gpurun_out/bench_$TAG.log').read()[-2000:])
PY
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_$TAG" -o t -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/prof_$TAG.log" 2>&1; echo "rocprof rc=$?"
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_$TAG/t_results.db" --steps 6 > "$R/gpurun_out/prof_$TAG.txt"; cat "$R/gpurun_out/prof_$TAG.txt"
