#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
cat > /tmp/vd.py <<'PY'
import torch, time
A = torch.randn(4096, 4096, dtype=torch.float64, device='cuda'); B = torch.randn(4096, 10048, dtype=torch.float64, device='cuda')
for _ in range(5): C = A @ B
torch.cuda.synchronize()
At = torch.randn(4096, 4096, dtype=torch.float64, device='cuda').t(); 
for _ in range(5): C = At @ B
torch.cuda.synchronize()
PY
export TMPDIR=/tmp; R="$GRAFT_REPO_ROOT"; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_vendor" -o t -- python /tmp/vd.py > "$R/gpurun_out/prof_vendor.log" 2>&1
python "$R/tools/prof_summary.py" "$R/gpurun_out/prof_vendor/t_results.db" --steps 5 2>/dev/null | head -12
python - <<'PY'
import sqlite3, glob, os
db = glob.glob(os.environ.get('GRAFT_REPO_ROOT','.') + '/gpurun_out/prof_vendor/t_results.db')[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
for t in tabs:
    if 'kernel_symbol' in t or 'kernel_dispatch' in t:
        cols = [r[1] for r in c.execute(f'pragma table_info({t})')]
        print(t, cols[:30])
for t in tabs:
    if 'kernel_symbol' in t:
        for r in c.execute(f'select * from {t}').fetchall()[:10]:
            print([str(x)[:300] for x in r])
PY
ls /opt/rocm/lib/rocblas/library | grep -i gfx950 | head; ls /opt/rocm/lib/hipblaslt/library | grep -i gfx950 | head
