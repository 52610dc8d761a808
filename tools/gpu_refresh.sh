#!/bin/bash
# refresh of the judged artefacts: GPU tests, kernel-trace profile of bench.py, secondary configs, vendor DGEMM reference
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r01b" -o t -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/prof_r01b.log" 2>&1; echo "rocprof rc=$?"
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_r01b/t_results.db" --steps 6 > "$R/gpurun_out/prof_r01b.txt"; head -40 "$R/gpurun_out/prof_r01b.txt"
cd "$R"
timeout 600 python tools/bench_c3.py > gpurun_out/c3.jsonl 2>gpurun_out/c3.err; cat gpurun_out/c3.jsonl
timeout 600 python tools/bench_train.py > gpurun_out/c4.jsonl 2>gpurun_out/c4.err; tail -3 gpurun_out/c4.jsonl
timeout 300 python - <<'PY'
import torch, time, json
torch.backends.cuda.matmul.allow_tf32 = False
A = torch.randn(4096, 4096, dtype=torch.float64, device='cuda'); B = torch.randn(4096, 10048, dtype=torch.float64, device='cuda')
for _ in range(3): C = A @ B
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): C = A @ B
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(json.dumps({'bench': 'vendor DGEMM (torch.matmul f64 -> rocBLAS/hipBLASLt) 4096x4096x10048 dense', 'ms': dt * 1e3, 'tflops': 2 * 4096 * 4096 * 10048 / dt * 1e-12}))
PY
