#!/bin/bash
# how long do workspace-sized hipMalloc / hipFree / hipMemset calls take on this box (gpmpc_append re-allocates its workspaces)
cat > /tmp/mt.py <<'PY'
import ctypes, time
hip = ctypes.CDLL('/opt/rocm/lib/libamdhip64.so')
def t(f):
    t0 = time.perf_counter(); r = f(); return (time.perf_counter() - t0) * 1e3, r
n = 6 * 8256 * 8256 * 8
for rep in range(3):
    ps = []
    for k in range(4):
        p = ctypes.c_void_p()
        ms, rc = t(lambda: hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n)))
        ps.append(p); print('rep %d hipMalloc %.2f GB: %.2f ms rc=%d' % (rep, n / 1e9, ms, rc))
    ms, rc = t(lambda: (hip.hipMemset(ps[0], 0, ctypes.c_size_t(n)), hip.hipDeviceSynchronize()))
    print('  memset+sync %.2f ms' % ms)
    for p in ps:
        ms, rc = t(lambda: hip.hipFree(p)); print('  hipFree %.2f ms' % ms)
PY
python /tmp/mt.py
