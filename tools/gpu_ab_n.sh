#!/bin/bash
# A/B of one environment switch at another problem size: tools/gpu_ab_n.sh N VAR "v1 v2 ..." [runs]
mkdir -p gpurun_out
N=$1; VAR=$2; VALS=$3; RUNS=${4:-2}
for v in $VALS; do for i in $(seq $RUNS); do
  env $VAR=$v timeout 300 python bench.py --N $N --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('N=$N $VAR=$v value %.0f  ms/step %.3f  factor %.3f  chain %.3f  vargemm %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm']))"
done; done
