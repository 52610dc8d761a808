#!/bin/bash
# A/B of one environment switch on the C2 bench line, no tests afterwards: tools/gpu_ab_notest.sh VAR "v1 v2 ..." [runs]
mkdir -p gpurun_out
VAR=$1; VALS=$2; RUNS=${3:-2}
for v in $VALS; do for i in $(seq $RUNS); do
  env $VAR=$v timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$VAR=$v value %.0f  ms/step %.3f  factor %.3f  chain %.3f  solve %.3f crosscov %.3f vargemm %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['solve'], p['crosscov'], p['vargemm']))"
done; done
