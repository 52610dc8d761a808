#!/bin/bash
# r04: hyper rows / jitters as kernel arguments; 16-row skipping inside the diagonal blocks of the variance product (A/B against
# a build without it, made on the box)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "variance_persistent or c2_full or behind_tail or jitter or synthetic or tank or car_model or training" 2>&1 | tail -3
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  value %.0f  ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f frac %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm'], j['roofline']['frac']))"
}
for rep in 1 2; do
  GPMPC_PARAM_KERNEL=0 run "skip, params copied      "
  run "skip, params by kernel   "
done
cd gp_mpc_amd/csrc && cp libgpmpc_hip.so /tmp/skip.so && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DGPMPC_VAR_DIAG_SKIP=0 gpmpc_api.hip -o libgpmpc_hip.so 2>/dev/null; cd ../..
for rep in 1 2; do
  run "no skip, params by kernel"
done
cp /tmp/skip.so gp_mpc_amd/csrc/libgpmpc_hip.so
for rep in 1 2; do
  run "skip, params by kernel   "
done
