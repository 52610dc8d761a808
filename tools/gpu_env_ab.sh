#!/bin/bash
# A/B of one environment switch on the C2 bench line: tools/gpu_env_ab.sh VAR "v1 v2 ..."   ("-" = unset)
cd "$GRAFT_REPO_ROOT"
for v in $2; do
  for i in 1 2; do
    if [ "$v" = "-" ]; then unset $1; else export $1=$v; fi
    timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1=$v  value %.0f  ms/step %.3f  factor %.3f  chain %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain']))"
  done
done
