#!/bin/bash
# first GPU pass: smoke, parity tests, bench, kernel-trace profile
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.log 2>&1
nproc >> gpurun_out/device.log
echo "== smoke" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -5 gpurun_out/bench.log
echo "== rocprof"
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r1" -o r1 -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/prof.log" 2>&1; echo "rocprof rc=$?"; tail -3 "$R/gpurun_out/prof.log"
ls -R "$R/gpurun_out/prof_r1" | head -20
