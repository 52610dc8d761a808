#!/bin/bash
# worker-launch cuts / worker counts of the chained factorisation: factor time per setting (one box)
mkdir -p gpurun_out
run() { env "$@" timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$* value %.0f  ms/step %.3f  factor %.3f  chain %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain']))"; }
run GPMPC_MAX_LAUNCHES=3
run GPMPC_NW2=64
run GPMPC_NW2=80
run GPMPC_NW2=128
run GPMPC_NW3=16
run GPMPC_NW3=24
run GPMPC_NW3=48
run GPMPC_CUT1=28 GPMPC_CUT2=48
run GPMPC_CUT1=36 GPMPC_CUT2=52
run GPMPC_CUT1=32 GPMPC_CUT2=52
run GPMPC_CUT1=32 GPMPC_CUT2=44
run GPMPC_MAX_LAUNCHES=2
run GPMPC_MAX_LAUNCHES=3
