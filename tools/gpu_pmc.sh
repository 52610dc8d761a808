#!/bin/bash
# PMC passes (separate from tracing-heavy options, as gpurun requires): HBM traffic + MFMA busy of the bench step
TAG=${1:-pmc}
R="$GRAFT_REPO_ROOT"; mkdir -p "$R/gpurun_out"; export TMPDIR=/tmp; cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$R/gpurun_out/pmc_${TAG}_$N" -o p -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$R/gpurun_out/pmc_${TAG}_$N.log" 2>&1
  echo "== $C rc=$?"
  python "$R/tools/pmc_summary.py" "$R/gpurun_out/pmc_${TAG}_$N/p_results.db" $2 > "$R/gpurun_out/pmc_${TAG}_$N.txt" 2>&1
  grep -A12 -E "gemm_f64_dma_kernel|gemm_f64_kernel<128|leaf64|crosscov|gram_kernel" "$R/gpurun_out/pmc_${TAG}_$N.txt" | head -60
done
