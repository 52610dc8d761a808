#!/bin/bash
R="$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; cd /tmp
python "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %8.0f  ms/step %.3f  factor %.3f  vargemm %.3f  %.1f TF' % (d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor'], d['phases_ms_per_step']['vargemm'], d['roofline']['achieved']))"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$R/gpurun_out/pmc_t" -o p -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import sqlite3
con=sqlite3.connect("$R/gpurun_out/pmc_t/p_results.db")
rows=con.execute("select kernel_name, grid_size_x, workgroup_size_x, value, dispatch_id from counters_collection where counter_name='FETCH_SIZE'").fetchall()
acc={}
for n,gx,wx,v,d in rows:
    if 'gemm_f64_dma_kernel' in n and gx//wx>2000: acc[d]=acc.get(d,0)+v
print('variance GEMM FETCH_SIZE per launch: %.3f GB raw (x2 corrected %.3f GB), launches %d' % (sum(acc.values())/len(acc)/1e6*1.024, 2*sum(acc.values())/len(acc)/1e6*1.024, len(acc)))
PY
