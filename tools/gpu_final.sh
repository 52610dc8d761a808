#!/bin/bash
# end-of-round refresh of the judged artefacts
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -c 2500 gpurun_out/bench_full.json
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r01c" -o t -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/prof_r01c.log" 2>&1; echo "rocprof rc=$?"
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_r01c/t_results.db" --steps 6 > "$R/gpurun_out/prof_r01c.txt"; head -14 "$R/gpurun_out/prof_r01c.txt"
cd "$R"
timeout 600 python tools/bench_c3.py > gpurun_out/c3.jsonl 2>gpurun_out/c3.err; cat gpurun_out/c3.jsonl | cut -c1-260
timeout 600 python tools/bench_train.py > gpurun_out/c4.jsonl 2>gpurun_out/c4.err; tail -2 gpurun_out/c4.jsonl | cut -c1-300
timeout 300 python tools/bench_append.py 2>/dev/null > gpurun_out/append.jsonl; cat gpurun_out/append.jsonl | cut -c1-200
