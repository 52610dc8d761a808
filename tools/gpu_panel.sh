#!/bin/bash
# panel / leaf micro-benchmarks + a short bench line
mkdir -p gpurun_out
{ tools/ubench/panel_dpp_bench; echo "--- leaf, readlane form"; tools/ubench/leaf_loop_bench_old; echo "--- leaf, DPP form"; tools/ubench/leaf_loop_bench; } > gpurun_out/panel_dpp.txt 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_dpp.json 2> gpurun_out/bench_dpp.err
cat gpurun_out/panel_dpp.txt; python - <<'PY'
import json
j = json.load(open('gpurun_out/bench_dpp.json'))
print(j['value'], j['ms_per_step'], j['phases_ms_per_step'], j.get('parity_vs_cpu'))
PY
tail -3 gpurun_out/bench_dpp.err
