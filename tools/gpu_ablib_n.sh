#!/bin/bash
# A/B of library builds at another problem size: tools/gpu_ablib_n.sh N lib1 lib2 ...
mkdir -p gpurun_out
N=$1; shift
cp gp_mpc_amd/csrc/libgpmpc_hip.so /tmp/orig.so
for round in 1 2; do for v in "$@"; do
  cp tools/ab_libs/$v.so gp_mpc_amd/csrc/libgpmpc_hip.so
  timeout 300 python bench.py --N $N --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('N=$N $v value %.0f  ms/step %.3f  factor %.3f  chain %.3f  vargemm %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm']))"
done; done
cp /tmp/orig.so gp_mpc_amd/csrc/libgpmpc_hip.so
