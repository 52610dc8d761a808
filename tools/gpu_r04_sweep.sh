#!/bin/bash
# r04: env-only re-tune after the write-through hand-offs: the chain's late polls, workers of the second / third launch
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
run() {
  timeout 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  C2 ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f' % (j['ms_per_step'], p['factor'], p['chain'], p['vargemm']))"
}
for rep in 1 2; do
  run "default          "
  for lp in 2 5 8; do GPMPC_LATE_POLLS=$lp run "LATE_POLLS=$lp     "; done
  for nw in 64 128; do GPMPC_NW2=$nw run "NW2=$nw          "; done
  for nw in 24 48; do GPMPC_NW3=$nw run "NW3=$nw           "; done
done
