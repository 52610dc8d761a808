#!/bin/bash
# round-3 check: GPU test tier, smoke, default bench line (C2 + secondary), launcher form with one GPU
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/gputests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/gputests.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open('gpurun_out/bench_r03.json'))
print('value %.0f ms/step %.3f' % (j['value'], j['ms_per_step']), j['phases_ms_per_step'])
print('roofline', j['roofline']['frac'], j['roofline']['traffic_source'])
print('secondary', json.dumps(j.get('secondary'))[:1200])
print('cpu', j.get('cpu_baseline', {}).get('value'), j.get('parity_vs_cpu'))
PY
tail -3 gpurun_out/bench_r03.err
