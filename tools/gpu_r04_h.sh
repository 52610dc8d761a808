#!/bin/bash
# r04: schedule that levels the XCDs first (tiles stay on their panel's XCD) -- time and fetched bytes against the dispatcher's order
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q -k "variance_persistent or c2_full_size_vs or behind_tail" 2>&1 | tail -3
run() {
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('$1  value %.0f  ms/step %.3f  factor %.3f  chain %.3f vargemm %.3f frac %.3f' % (j['value'], j['ms_per_step'], p['factor'], p['chain'], p['vargemm'], j['roofline']['frac']))"
}
for rep in 1 2 3; do
  GPMPC_VARGEMM_PERSIST=0 run "dispatcher order "
  GPMPC_VARGEMM_PERSIST=1 run "static schedule  "
done
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  GPMPC_VERBOSE=1 timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$R/gpurun_out/pmc_r04_$N" -o p -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$R/gpurun_out/pmc_r04_$N.log" 2>&1
  echo "== $C rc=$?"
  python "$R/tools/pmc_summary.py" "$R/gpurun_out/pmc_r04_$N/p_results.db" > "$R/gpurun_out/r04_pmc_$N.txt" 2>&1; grep -A4 "vargemm_persist" "$R/gpurun_out/r04_pmc_$N.txt" | head -6
done
grep "schedule" "$R/gpurun_out/pmc_r04_FETCH_SIZE.log" | head -2
