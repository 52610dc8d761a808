#!/bin/bash
# two processes sharing ONE GPU: both run the C2 bench loop; hand-off time-outs fall back per factorisation, results stay right
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for i in 1 2; do
  (timeout 300 python bench.py --steps 300 --warmup 3 --no-cpu-baseline > gpurun_out/two_$i.json 2> gpurun_out/two_$i.err) &
done
wait
for i in 1 2; do
  python -c "
import json
j=json.loads(open('gpurun_out/two_$i.json').read().strip().splitlines()[-1]); print('proc $i: value %.0f ms/step %.3f parity' % (j['value'], j['ms_per_step']), j.get('parity_vs_cpu'))"
  echo "  time-out messages: $(grep -c 'timed out' gpurun_out/two_$i.err)"
done
