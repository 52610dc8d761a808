#!/bin/bash
# Refresh of the judged artefacts on ONE box: tools/gpu_refresh.sh <tag> [parts]      (e.g. `gpurun -- tools/gpu_refresh.sh r05 "tests bench trace pmc"`)
#   tests  GPU tier + smoke()                          -> gpurun_out/<tag>_gpu_tests.txt
#   bench  bench.py C2 (CPU baseline, secondary), --config C3, --config C4 -> <tag>_bench.json, <tag>_bench_c3.json, <tag>_bench_c4.json
#   trace  rocprofv3 --kernel-trace --stats of the C2 / C3 / C4 commands, C2 step timeline, chain + worker stamps
#          -> <tag>_kernel_trace_bench{,_c3,_c4}.txt, <tag>_step_timeline.txt, <tag>_chain_trace.txt, <tag>_worker_trace.txt
#   pmc    separate --pmc passes of the C2 step (fetched / written bytes, matrix-pipe duty) + of one C3 exact-moment step
#          (VALU / LDS counters of the pair sums) -> <tag>_pmc_*.txt, <tag>_traffic.json
#   soak   1000 consecutive C2 steps, hand-off time-outs counted -> <tag>_soak.txt
# Copy what is to be judged from gpurun_out/ to profiles/.
TAG=${1:-rXX}; PARTS=${2:-"tests bench trace"}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/${TAG}_gpu_tests.txt 2>&1; grep -E "passed|failed|error" $O/${TAG}_gpu_tests.txt | tail -3
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
fi
if has bench; then
  timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; python - "$O/${TAG}_bench.json" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('C2 value %.0f ms/step %.3f (all brackets %.3f) vargemm frac %.3f cholesky frac %.3f phases %s' % (j['value'], j['ms_per_step'], j['ms_per_step_all_brackets'], j['roofline']['frac'], j['cholesky']['frac'], {k: round(v, 3) for k, v in j['phases_ms_per_step'].items()}))
print('cpu', j.get('cpu_baseline', {}).get('value'), 'parity_vs_oracle', j.get('parity_vs_oracle'), 'timeouts', j.get('handoff_timeouts'))
print('secondary c3 ms', j['secondary']['c3']['ms_per_step'], 'em', j['secondary']['c3']['phases_ms_per_step'].get('em'), 'c4 restarts/s', j['secondary']['c4']['restarts_per_s'], j['secondary']['c4'].get('shard_check'))
PY
  timeout 600 python bench.py --config C3 > $O/${TAG}_bench_c3.json 2> $O/${TAG}_bench_c3.err; cut -c1-300 $O/${TAG}_bench_c3.json
  timeout 600 python bench.py --config C4 > $O/${TAG}_bench_c4.json 2> $O/${TAG}_bench_c4.err; cut -c1-300 $O/${TAG}_bench_c4.json
fi
if has trace; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_$TAG" -o t -- python "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary > "$O/prof_$TAG.log" 2>&1; echo "rocprof C2 rc=$?"
  python "$R/tools/prof_summary.py" "$O/prof_$TAG/t_results.db" --steps 6 > "$O/${TAG}_kernel_trace_bench.txt"; head -12 "$O/${TAG}_kernel_trace_bench.txt"
  python "$R/tools/step_timeline.py" "$O/prof_$TAG/t_results.db" > "$O/${TAG}_step_timeline.txt" 2>&1; rm -rf "$O/prof_$TAG"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_${TAG}c3" -o t -- python "$R/bench.py" --config C3 --steps 2 --warmup 1 > "$O/prof_${TAG}c3.log" 2>&1; echo "rocprof C3 rc=$?"
  python "$R/tools/prof_summary.py" "$O/prof_${TAG}c3/t_results.db" --steps 3 > "$O/${TAG}_kernel_trace_bench_c3.txt"; head -8 "$O/${TAG}_kernel_trace_bench_c3.txt"; rm -rf "$O/prof_${TAG}c3"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_${TAG}c4" -o t -- python "$R/bench.py" --config C4 --steps 1 --warmup 1 > "$O/prof_${TAG}c4.log" 2>&1; echo "rocprof C4 rc=$?"
  python "$R/tools/prof_summary.py" "$O/prof_${TAG}c4/t_results.db" --steps 2 > "$O/${TAG}_kernel_trace_bench_c4.txt"; head -8 "$O/${TAG}_kernel_trace_bench_c4.txt"; rm -rf "$O/prof_${TAG}c4"
  cd "$R"
  GPMPC_CHAIN_TRACE=$O/ct_$TAG.bin timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
  python tools/chain_trace.py $O/ct_$TAG.bin 64 > $O/${TAG}_chain_trace.txt 2>&1; head -13 $O/${TAG}_chain_trace.txt | tail -5
  python tools/worker_trace.py $O/ct_$TAG.bin 64 2>&1 | grep -v "^ *[0-9]*a .*-7[0-9][0-9][0-9][0-9][0-9][0-9]" > $O/${TAG}_worker_trace.txt; rm -f $O/ct_$TAG.bin
fi
if has pmc; then
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-24)
    timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$O/pmc_${TAG}_$N" -o p -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > "$O/pmc_${TAG}_$N.log" 2>&1
    echo "== $C rc=$?"
    python "$R/tools/pmc_summary.py" "$O/pmc_${TAG}_$N/p_results.db" > "$O/${TAG}_pmc_$N.txt" 2>&1; grep -A4 "vargemm_persist" "$O/${TAG}_pmc_$N.txt" | head -6
  done
  python "$R/tools/traffic_json.py" "$O/pmc_${TAG}_FETCH_SIZE/p_results.db" "$O/pmc_${TAG}_WRITE_SIZE/p_results.db" "$O/pmc_${TAG}_SQ_VALU_MFMA_BUSY_CYCLES/p_results.db" "$TAG" > "$O/${TAG}_traffic.json" 2>"$O/${TAG}_traffic.err"; head -c 600 "$O/${TAG}_traffic.json"; echo
  rm -rf $O/pmc_${TAG}_FETCH_SIZE $O/pmc_${TAG}_WRITE_SIZE $O/pmc_${TAG}_SQ_VALU_MFMA_BUSY_CYCLES
  cat > /tmp/emv.py <<'PY'
import sys, os
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from gp_mpc_amd.synthetic import synthetic_problem
from gp_mpc_amd._lib import Handle, get_lib
p = synthetic_problem(8192, 8, 6, 4, seed=1234, sn=1e-2)
h = Handle(get_lib(), p['X'], p['Y'])
h.fit(p['hyper'], want_invK=True); h.synchronize()
for _ in range(3): h.predict('EM', p['Z'][:1], p['Sigma'][:1])
h.synchronize()
PY
  for C in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-24)
    timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$O/pmc_em_$N" -o p -- python /tmp/emv.py > "$O/pmc_em_$N.log" 2>&1
    echo "== EM $C rc=$?"
    python "$R/tools/pmc_summary.py" "$O/pmc_em_$N/p_results.db" > "$O/${TAG}_pmc_em_$N.txt" 2>&1; grep -E -A5 "em_pair|em_diag" "$O/${TAG}_pmc_em_$N.txt" | head -14; rm -rf "$O/pmc_em_$N"
  done
  cd "$R"
fi
if has soak; then
  timeout 300 python bench.py --steps 1000 --warmup 3 --no-cpu-baseline --no-secondary 2>$O/soak_err.log | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('soak 1000 steps: value %8.0f  ms/step %.3f  factor %.3f vargemm %.3f  handoff_timeouts %s' % (d['value'], d['ms_per_step'], d['phases_ms_per_step']['factor'], d['phases_ms_per_step']['vargemm'], d.get('handoff_timeouts')))" | tee $O/${TAG}_soak.txt
  echo "time-out messages: $(grep -c 'timed out' $O/soak_err.log)" | tee -a $O/${TAG}_soak.txt
fi
