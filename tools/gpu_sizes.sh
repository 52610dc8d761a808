#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for N in 512 1024 1536 2048 2560 3072 3584 4096; do
  GPMPC_VERBOSE=1 timeout 200 python bench.py --N $N --B 2000 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/tmp/err.txt | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('N=$N ms/step %.3f  gram %.3f factor %.3f  chain %.3f  N^3/3/chain = %.1f TF' % (j['ms_per_step'], p['gram'], p['factor'], p['chain'], ($N**3/3)/(p['chain']*1e-3)/1e12))"
  grep "gpmpc: factor" /tmp/err.txt | sort | uniq -c | head -2
done
