#!/bin/bash
# r04: re-sweep of the fit's tunables with the prediction overlapped behind the tail (they were set in r03 without it)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() {
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  value %.0f  ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f crosscov %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm'], p['crosscov']))"
}
run "default              "
GPMPC_T64=128 run "T64=128              "
GPMPC_T64=512 run "T64=512              "
GPMPC_T64=1024 run "T64=1024             "
GPMPC_T64_STAGES=2 run "T64 2 images         "
GPMPC_T64_STAGES=3 run "T64 3 images         "
GPMPC_CROSSCOV_WGS=128 run "crosscov 128 wgs     "
GPMPC_CROSSCOV_WGS=512 run "crosscov 512 wgs     "
GPMPC_CROSSCOV_WGS=0 run "crosscov unthrottled "
GPMPC_NW2=64 run "second launch 64     "
GPMPC_NW2=128 run "second launch 128    "
GPMPC_NW3=16 run "third launch 16      "
GPMPC_NW3=64 run "third launch 64      "
GPMPC_TRTRI_QUEUE=0 run "I_i on the product q "
GPMPC_ALPHA_SIDE=0 run "alpha on main queue  "
run "default again        "
