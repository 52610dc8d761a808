#!/bin/bash
# r04: threshold for 64-row GEMM tiles (GPMPC_T64: blocks from which 64-row tiles are taken instead of 32-row ones), all three configs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  C2 ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f' % (j['ms_per_step'], p['factor'], p['chain'], p['vargemm']))"
}
c3() { timeout 300 python bench.py --config C3 --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 C3 ms/step %.1f' % j['ms_per_step'], {k: round(v,2) for k,v in j.get('phases_ms_per_step',{}).items()})"; }
c4() { timeout 300 python bench.py --config C4 --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 C4 restarts/s %.1f' % j['value'])"; }
for rep in 1 2 3; do
  GPMPC_T64=256 run "T64=256"
  GPMPC_T64=512 run "T64=512"
  GPMPC_T64=384 run "T64=384"
done
GPMPC_T64=256 c3 "T64=256"; GPMPC_T64=512 c3 "T64=512"
GPMPC_T64=256 c4 "T64=256"; GPMPC_T64=512 c4 "T64=512"
