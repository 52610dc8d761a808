#!/usr/bin/env python3
"""bench.py -- headline benchmark of the GP hot path (BASELINE.json metric) on MI355X.

Workload (config.workload "C2"): BASELINE.json configs[1] -- single-output SE-ARD GP, N=4096, d=6,
fp64: one STEP = K build + Cholesky (+ L^-1, alpha) at fixed hyper-parameters + 10 000 mean+variance
predictions, everything through the C ABI of libgpmpc_hip.so with X, Y, Z and the outputs resident
in HBM (device pointer mode); since r06 as ONE call, gpmpc_fit_predict_mean_var (bitwise gpmpc_fit +
gpmpc_predict_mean_var, which --two-calls times instead; the line carries the other form's ms_per_step too).  `value` = predictions/s over whole steps (fit included), summed over
ranks.  With --gpus N every rank runs its own model / test batch on its own GPU (independent GP
objects shard trivially, no data-path collective): weak scaling.  `python bench.py --gpus N` launches its N ranks
itself (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0 prints the line); under
torch.distributed.run the ranks it finds in the environment are used.  For N > 1 the line also carries `restart_shard`:
the ONE part of the path that shards (BASELINE config C4) -- 64 seeded restarts of the NLL minimisation, restart r on
rank r mod N, ONE ncclAllGather of the (NLL, theta) table inside gpmpc_train_multistart -- as restarts/s (strong
scaling) with `rccl_ranks` = ncclCommCount of the communicator that carried it.  For N = 1 the line carries `secondary`
(after the timed C2 region): two steps of C3 (the three methods in lock-step, gpmpc_rollout_multi; `rollout_hbm`), one of C4 (with a
flop-counted `roofline`), `c5` (Nt = 30 shooting nodes per call: value + Jacobian + TA covariance, 50 calls, at C3 size and at the
reference's car-model size) and `b1` (B = 1 streaming predictions on the fitted C2 model), each with its HBM roofline.

--config C3 (secondary, same JSON shape; the default and the headline stay C2): BASELINE.json configs[2] --
6-output GP, N=8192, d=8: one STEP = fit of all outputs with K^-1 + a 30-step uncertainty propagation with each
of 'ME', 'TA', 'EM' (gpmpc_rollout); `value` = propagation steps/s over whole steps (fit included).
--config C4: BASELINE.json configs[3] -- hyper-parameter training at N=4096, d=6: one STEP = 64 seeded restarts (4 L-BFGS
iterations each) of `gpmpc_train_multistart`, restart r on rank r mod world, the (NLL, theta) table exchanged by ONE
ncclAllGather over an RCCL communicator the library creates (strong scaling: the 64 restarts are fixed); `value` = restarts/s.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (variance GEMM V = L^-1 Ks with fused column sums of squares):
                  algorithmic flops per launch N(N+1) B / HIP-event time of that launch, vs the fp64
                  MFMA peak (78.6 TFLOP/s spec; the rate of a pure MFMA loop measured at library load is reported beside it).
  cholesky     -- the factorisation alone against the same peak: N^3/3 flops / HIP-event time of the persistent
                  chain kernel, which ends with the last leaf, i.e. when L is complete (the fused L^-1 is extra).
  cpu_baseline -- the oracle's reference-formulation CPU path (numpy/LAPACK, same box, rank 0, N=1):
                  one full fit + 1000 of the 10 000 predictions (prediction time scaled x10); `fair` = the same
                  with triangular solves instead of the reference's general LU solves; BLAS threads / version.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _oracle():
    """The CPU restatement under oracle/ -- imported by the cpu_baseline / parity legs ONLY (after the timed regions): the
    product legs take their inputs from gp_mpc_amd.synthetic and never see oracle/ on sys.path before this call."""
    op = os.path.join(ROOT, 'oracle')
    if op not in sys.path:
        sys.path.insert(0, op)
    import gp_oracle
    return gp_oracle

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet, dense fp64 matrix (not in the local guides)


def cpu_baseline(p, B):
    """Oracle ("port") timed on the host cores: the reference formulation exactly --
    expanded-form K (optimize.py:303-319), np.linalg.cholesky, LU np.linalg.solve for alpha and for
    v = L^-1 ks (gp_class.py:377-380), var = sf^2 - v^T v."""
    import numpy as np
    go = _oracle()
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    d = X.shape[1]
    nsub = min(1000, B)
    t0 = time.perf_counter()
    K = go.gram(X, H[0, :d], H[0, d] ** 2, H[0, d + 1] ** 2)
    L, _ = go.chol_jitter(K)
    t_kl = time.perf_counter() - t0
    t0 = time.perf_counter()
    alpha = go.alpha_from_chol(L, Y[:, 0])
    t_fit = t_kl + (time.perf_counter() - t0)
    t0 = time.perf_counter()
    ks = go.cov_se_ard(X, Z[:nsub], H[0, :d], H[0, d] ** 2)
    mean = ks.T @ alpha
    v = np.linalg.solve(L, ks)
    var = H[0, d] ** 2 - np.sum(v * v, axis=0)
    t_pred = (time.perf_counter() - t0) * (B / nsub)
    # fair CPU variant: same K and Cholesky, triangular solves where the reference uses general LU
    from scipy.linalg import solve_triangular
    t0 = time.perf_counter()
    alpha2 = solve_triangular(L, solve_triangular(L, Y[:, 0], lower=True), lower=True, trans='T')
    t_fit_fair = t_kl + (time.perf_counter() - t0)
    t0 = time.perf_counter()
    ks2 = go.cov_se_ard(X, Z[:nsub], H[0, :d], H[0, d] ** 2)
    v2 = solve_triangular(L, ks2, lower=True)
    var2 = H[0, d] ** 2 - np.sum(v2 * v2, axis=0)
    mean2 = ks2.T @ alpha2
    t_pred_fair = (time.perf_counter() - t0) * (B / nsub)
    blas = {}
    try:
        from threadpoolctl import threadpool_info
        for lib_ in threadpool_info():
            if lib_.get('user_api') == 'blas':
                blas = {'blas': lib_.get('internal_api'), 'version': lib_.get('version'), 'threads': lib_.get('num_threads')}
    except Exception:
        pass
    threads = blas.get('threads') or os.cpu_count()
    return dict(value=B / (t_fit + t_pred), unit='predictions/s', cores=threads, kind='port',
                sample=f'1 full fit (N={X.shape[0]}) + {nsub} of {B} predictions, prediction time scaled x{B // nsub}; '
                       f'fit {t_fit:.2f} s, predictions {t_pred:.2f} s (scaled); numpy {np.__version__}',
                fit_s=t_fit, predict_s=t_pred, host_cpus=os.cpu_count(), blas=blas,
                fair={'value': B / (t_fit_fair + t_pred_fair), 'unit': 'predictions/s', 'predict_s': t_pred_fair,
                      'note': 'solve_triangular for alpha and v = L^-1 ks instead of np.linalg.solve (LU)'}), mean, var, \
        np.abs(ks).T @ np.abs(alpha)


def parity_report(gm, gv, cmean, cvar, mscale, sf2):
    """GPU against the CPU reference-formulation path on the same inputs: raw relative errors (north_star's '1e-10 rel') next
    to the condition-scaled measures the parity tests gate on (tests/parity_cases.py)."""
    import numpy as np
    dm, dv = np.abs(gm - cmean), np.abs(gv - cvar)
    return {'mean_maxabs': float(dm.max()), 'mean_rel_to_max': float(dm.max() / np.abs(cmean).max()),
            'mean_max_pointwise_rel': float((dm / np.maximum(np.abs(cmean), 1e-300)).max()),
            'mean_scaled_sum_abs_ks_alpha': float((dm / mscale).max()),
            'mean_pointwise_floored_1e-2_max': float((dm / np.maximum(np.abs(cmean), 1e-2 * np.abs(cmean).max())).max()),
            'var_maxabs_over_sf2': float(dv.max() / sf2), 'var_max_pointwise_rel': float((dv / np.abs(cvar)).max()),
            'points': int(len(cmean))}


def oracle_parity(p, gm, gv, nsub):
    """GPU against the oracle the parity tests use (oracle/gp_oracle.py `fit` + `mean_var_jac`: direct-difference ks,
    Cholesky, TRIANGULAR solves -- gp_functions.py:114-126 restated) on the first `nsub` test points of the same inputs."""
    import numpy as np
    go = _oracle()
    X, Y, H, Z = p['X'], p['Y'], p['hyper'], p['Z']
    d = X.shape[1]
    t0 = time.perf_counter()
    o = go.fit(X, Y, H, want_invK=False)
    om, ov, _ = go.mean_var_jac(Z[:nsub], X, H, o['alpha'], o['chol'], False)
    secs = time.perf_counter() - t0
    ks = go.cov_se_ard_direct(X, Z[:nsub], H[0, :d], H[0, d] ** 2)
    mscale = np.abs(ks).T @ np.abs(o['alpha'][0])
    rep = parity_report(gm[:nsub], gv[:nsub], om[:, 0], ov[:, 0], mscale, float(H[0, d] ** 2))
    rep['oracle'] = 'oracle/gp_oracle.py fit + mean_var_jac (triangular solves), the checker of tests/parity_cases.py'
    rep['oracle_seconds'] = secs
    return rep


_REAL_STDOUT = None
_T_START = time.time()


def _own_stdout():
    """The contract is ONE JSON line on stdout (rank 0).  Libraries in the process write there too -- RCCL prints a version
    banner from C stdio, flushed at exit, i.e. AFTER a Python print -- so file descriptor 1 is pointed at stderr for
    everything else and the line goes to a private duplicate of the original stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)


def _emit(obj):
    _REAL_STDOUT.write(json.dumps(obj) + '\n')
    _REAL_STDOUT.flush()


def _launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, one process per GPU, and pass rank 0's JSON
    line through.  (The driver's torch.distributed.run form sets WORLD_SIZE itself and never reaches this.)"""
    import socket
    import subprocess
    import tempfile
    last = None
    for attempt in range(3):                # (the free port is found by bind-and-close: another process may take it in between)
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        procs = []
        out0 = tempfile.TemporaryFile()     # rank 0's stdout (a file: nobody has to drain a pipe while we poll)
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else sys.stderr, stderr=sys.stderr))
        # poll every rank: the first one to exit non-zero ends the job at once (a rank that dies before the rendezvous would
        # otherwise leave the others in init_process_group for torch's full time-out)
        rcs = [None] * n
        failed = None
        t_done0 = None
        while any(rc is None for rc in rcs):
            for r, q in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = q.poll()
                    if rcs[r] not in (None, 0) and failed is None:
                        failed = r
            if failed is not None:
                break
            if rcs[0] == 0 and t_done0 is None:
                t_done0 = time.time()       # rank 0 is done: the others leave their last barrier within moments, or are stuck
            if t_done0 is not None and time.time() - t_done0 > 120.0:
                break
            time.sleep(0.05)
        for r, q in enumerate(procs):
            if rcs[r] is None:
                q.kill()                    # (this very process, by its handle)
                q.wait()
                rcs[r] = 'killed' if failed is None else 'killed after rank %d failed' % failed
        out0.seek(0)
        out = out0.read()
        out0.close()
        last = rcs
        if not any(rcs):
            break
        if attempt == 2 or time.time() - _T_START > 60.0:      # (a rendezvous that cannot bind fails within seconds; anything later is a real failure)
            raise SystemExit('bench.py: rank exit codes %s' % rcs)
        sys.stderr.write('bench.py: ranks exited with %s within the first minute (port %d taken?); launching again\n' % (rcs, port))
    lines = [ln for ln in out.decode().splitlines() if ln.strip()]
    _REAL_STDOUT.write(lines[-1] + '\n')
    _REAL_STDOUT.flush()


class Ranks:
    """This process's place in the job: rank / world from the environment, the torch device, the process group
    (nccl = RCCL on a GPU box; gloo when tests/ points GPMPC_BENCH_LIB at the emulated library on a CPU box)."""

    def __init__(self):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.test_lib = os.environ.get('GPMPC_BENCH_LIB')       # set by tests/test_bench_launcher.py only
        self.gpu = torch.cuda.is_available() and not self.test_lib
        if not self.gpu and not self.test_lib:
            raise SystemExit('bench.py needs a GPU (torch.cuda.is_available() is False)')
        if self.gpu:
            torch.cuda.set_device(self.local_rank)
        self.dev_index = self.local_rank if self.gpu else 0      # (the emulated test library has one device)
        self.device = torch.device('cuda', self.local_rank) if self.gpu else torch.device('cpu')
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if self.gpu:
                dist.init_process_group('nccl', rank=self.rank, world_size=self.world, device_id=self.device)
            else:
                dist.init_process_group('gloo', rank=self.rank, world_size=self.world)
            self.dist = dist
        if self.test_lib:
            from gp_mpc_amd._lib import GpmpcLib
            self.lib = GpmpcLib(self.test_lib)
        else:
            from gp_mpc_amd._lib import get_lib
            self.lib = get_lib()

    def sync(self, h=None):
        if h is not None:
            h.synchronize()
        if self.gpu:
            self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed(rk, h, step, steps, warmup, profile=True, phases=None):
    """`warmup` untimed steps, then exactly `steps` steps between (barrier + synchronize) pairs; max over the ranks.
    profile: HIP-event brackets inside the library during the timed steps -- of the phases named, or of all."""
    for _ in range(warmup):
        step()
    rk.sync(h)
    if profile:
        h.profile_enable(True, phases)
        h.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    rk.sync(h)
    elapsed = time.perf_counter() - t0
    prof = None
    if profile:
        h.profile_enable(False)
        prof = h.profile_read(reset=True)
    return rk.max_over_ranks(elapsed), prof


def run_c3(rk, steps, warmup, N=8192):
    """BASELINE config C3: 6-output GP, N = 8192, d = 8; step = fit (with K^-1) + 30-step ME / TA / EM propagation."""
    import numpy as np
    from gp_mpc_amd._lib import Handle
    from gp_mpc_amd.synthetic import synthetic_problem
    d, Ny, T = 8, 6, 30
    p = synthetic_problem(N, d, Ny, T, seed=1234 + rk.rank, sn=1e-2)
    h = Handle(rk.lib, p['X'], p['Y'], device=rk.dev_index)
    hyper = np.ascontiguousarray(p['hyper'])
    x0, U = p['Z'][0, :Ny], p['Z'][:T, Ny:]
    S0 = np.eye(d) * 1e-6
    S0[:Ny, :Ny] = np.diag(hyper[:, d + 1] ** 2)
    z0 = np.concatenate([x0, U[0]])
    res = {}

    def step():
        h.fit(hyper, want_invK=True)
        # the three methods from the same start (r06, gpmpc_rollout_multi: one pass over L^-1 per time step for 'ME' and 'TA'
        # together on the main queue, the exact moments' whole horizon next to them on a second queue; r01-r05 called
        # gpmpc_rollout once per method.  Letting the roll-out call form K^-1 next to the 'ME' / 'TA' steps instead of inside
        # the fit measured slower: 111.7 against 108.0 ms, profiles/r06_rollout_overlap_ab.txt)
        mm, cc = h.rollout_multi(['ME', 'TA', 'EM'], z0, U, S0)
        for i, m in enumerate(('ME', 'TA', 'EM')):
            res[m] = (mm[i], cc[i])

    elapsed, prof = timed(rk, h, step, steps, warmup)
    # the roll-outs alone (phase brackets inside gpmpc_rollout are per launch; time them as whole calls; K^-1 is there by now)
    t_roll = {}
    for m in ('ME', 'TA', 'EM'):
        h.synchronize()
        t1 = time.perf_counter()
        h.rollout(m, z0, U, S0)
        t_roll[m] = (time.perf_counter() - t1) * 1e3
    for key, ms_ in (('ME+TA lock-step', ['ME', 'TA']), ('ME+TA+EM lock-step', ['ME', 'TA', 'EM'])):
        h.rollout_multi(ms_, z0, U, S0)
        h.synchronize()
        t1 = time.perf_counter()
        h.rollout_multi(ms_, z0, U, S0)
        t_roll[key] = (time.perf_counter() - t1) * 1e3
    fac_ms = prof['factor'][0] / max(prof['factor'][1], 1)
    flops = Ny * 2.0 * N ** 3 / 3.0
    em_ms, em_n = prof['em']
    # exact-moment pair sums: entries (one fp64 exp each) per input = 15 full N x N pair matrices + 6 lower triangles in 64-tiles;
    # bound by VALU issue: 14.8 instructions per entry (PMC, profiles/r05_pmc_em_*) at one wave instruction per 4 cycles and SIMD
    Np_, P_off, P_diag = (N + 63) // 64 * 64, Ny * (Ny - 1) // 2, Ny
    em_entries = P_off * float(Np_) ** 2 + P_diag * (Np_ // 64) * (Np_ // 64 + 1) / 2 * 4096.0
    EM_INSTR_PER_ENTRY, SIMDS, CLOCK = 14.8, 1024, 2.4e9
    em_peak = SIMDS * 64 * CLOCK / (4.0 * EM_INSTR_PER_ENTRY) * 1e-9          # Gentries/s the VALU can issue
    world = rk.world
    out = {
        'metric': 'GP moment-matching propagation steps/sec, N=8192 Ny=6 d=8 fp64 (fit + 30-step ME/TA/EM per step)',
        'rollout_api': 'gpmpc_rollout_multi([ME, TA, EM]): ME + TA as one batch per time step on the main queue, the EM horizon next to them on a second queue',
        'value': world * 3 * T * steps / elapsed, 'unit': 'propagation steps/s', 'n_gpus': world, 'steps': steps,
        'warmup': warmup, 'ms_per_step': elapsed / steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'C3: 6-output SE-ARD GP fit (K build + Cholesky + L^-1 + K^-1) + 30-step ME/TA/EM propagation',
                   'N': N, 'd': d, 'Ny': Ny, 'horizon': T, 'parallelism': f'independent GP per GPU x{world}'},
        'roofline': {'kernel': 'factorisation: two-level blocked Cholesky + pipelined triangular inverse (chain kernel + MFMA GEMM launches)',
                     'bound': 'mfma', 'achieved': flops / (fac_ms * 1e-3) * 1e-12 if fac_ms > 0 else 0.0,
                     'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': flops / (fac_ms * 1e-3) * 1e-12 / FP64_MFMA_PEAK_TFLOPS if fac_ms > 0 else 0.0,
                     'traffic': None, 'avg_launch_ms': fac_ms, 'launches': prof['factor'][1],
                     'note': '2 N^3 / 3 flops per output (potrf + trtri), HIP events around the whole factor phase'},
        'em_pair_kernels': {'bound': 'valu', 'achieved': em_entries / (em_ms / max(em_n, 1) * 1e-3) * 1e-9 if em_ms > 0 else 0.0,
                            'peak': em_peak, 'unit': 'Gentry/s', 'frac': (em_entries / (em_ms / max(em_n, 1) * 1e-3) * 1e-9 / em_peak) if em_ms > 0 else 0.0,
                            'avg_launch_ms': em_ms / max(em_n, 1), 'launches': em_n, 'entries_per_input': em_entries,
                            'note': 'one EM evaluation (operands, both pair-sum launches, finish) per launch figure; VALU-issue floor = 14.8 instructions per entry '
                                    '(rocprofv3 --pmc SQ_INSTS_VALU, profiles/r05_pmc_em_*) x 4 cycles per wave instruction on 1024 SIMDs at 2.4 GHz; HBM is at < 15 % here'},
        'phases_ms_per_step': {k: v[0] / steps for k, v in prof.items() if v[1] > 0},
        'rollout_ms_per_call': t_roll,
        'finite': bool(all(np.all(np.isfinite(r[0])) and np.all(np.isfinite(r[1])) for r in res.values())),
        'device': rk.lib.device_name(rk.dev_index),
    }
    # ME / TA roll-outs stream the six lower triangles of L^-1 once per time step (var_small_kernel): HBM-bound
    linv_bytes = Ny * 4.0 * N * (N + 1)
    out['rollout_hbm'] = {m: {'GBps': linv_bytes * T / (t_roll[m] * 1e-3) * 1e-9, 'frac_of_8TBps': linv_bytes * T / (t_roll[m] * 1e-3) / 8e12}
                          for m in ('ME', 'TA', 'ME+TA lock-step')}
    out['rollout_hbm']['note'] = ('algorithmic bytes per time step = the six lower triangles of L^-1 (%.2f GB), streamed once per step whether one or '
                                  'two trajectories ride on it; whole calls incl. the feed / cross-covariance / finish launches and the final copy' % (linv_bytes * 1e-9))
    out['c5'] = run_c5(h, p, N, Ny, d)
    h.close()
    return out


def run_c5(h, p, N, Ny, d, Nt=30, calls=50):
    """BASELINE config C5, the pattern an IPOPT iteration drives through the Callback: all Nt = 30 shooting nodes in one call,
    value + mean Jacobian + 'TA' covariance (gpmpc_predict_jac), `calls` calls, host arrays in and out (what casadi hands
    over).  HBM-bound: the lower triangles of L^-1 are streamed once per call for all nodes."""
    import numpy as np
    Z, Sg = p['Z'][:Nt], p['Sigma'][:Nt]
    h.predict_jac('TA', Z, Sg)
    h.profile_enable(True)
    h.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(calls):
        m, c, J = h.predict_jac('TA', Z, Sg)
    dt = (time.perf_counter() - t0) / calls
    h.profile_enable(False)
    prof = h.profile_read(reset=True)
    vg_ms = prof['vargemm'][0] / max(prof['vargemm'][1], 1)
    bytes_call = Ny * 4.0 * N * (N + 1)                  # lower triangle of L^-1 per output (+ 8 N d + 8 N, negligible)
    return {'workload': 'C5 pattern: Nt=%d nodes per call, value + J + TA covariance (gpmpc_predict_jac), %d calls, N=%d Ny=%d d=%d' % (Nt, calls, N, Ny, d),
            'ms_per_call': dt * 1e3, 'node_evals_per_s': Nt / dt, 'finite': bool(np.all(np.isfinite(c)) and np.all(np.isfinite(J))),
            'phases_ms_per_call': {k: v[0] / calls for k, v in prof.items() if v[1] > 0},
            'roofline': {'kernel': 'variance product for <= 32 columns (gemm_f64_dma_kernel<64,32,...>: L^-1 streamed once)', 'bound': 'hbm',
                         'achieved': bytes_call / (vg_ms * 1e-3) * 1e-9 if vg_ms > 0 else 0.0, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': bytes_call / (vg_ms * 1e-3) / 8e12 if vg_ms > 0 else 0.0, 'avg_launch_ms': vg_ms,
                         'algorithmic_bytes_per_call': bytes_call, 'whole_call_GBps': bytes_call / dt * 1e-9, 'traffic': None}}


def run_c5_small(rk, N=200, Ny=3, Nu=2, Nt=30, calls=50):
    """The same pattern at the size of the reference's own car model (N = 200, Ny = 3, d = 5; car_example.py:163-168,198):
    launch- and copy-latency-bound, reported as calls/s."""
    from gp_mpc_amd._lib import Handle
    from gp_mpc_amd.synthetic import synthetic_problem
    d = Ny + Nu
    p = synthetic_problem(N, d, Ny, Nt, seed=1234, sn=1e-2)
    h = Handle(rk.lib, p['X'], p['Y'], device=rk.dev_index)
    h.fit(p['hyper'])
    h.predict_jac('TA', p['Z'], p['Sigma'])
    t0 = time.perf_counter()
    for _ in range(calls):
        h.predict_jac('TA', p['Z'], p['Sigma'])
    dt = (time.perf_counter() - t0) / calls
    h.close()
    return {'workload': 'C5 pattern at the car model size: N=%d Ny=%d d=%d, Nt=%d nodes per call' % (N, Ny, d, Nt), 'ms_per_call': dt * 1e3,
            'node_evals_per_s': Nt / dt, 'bound': 'launch + copy latency (three launches, one upload, one download per call)'}


def run_b1(rk, h, N, d, calls=200):
    """SURVEY 8(d) 'predict, B=1 streaming (MPC-node pattern)': one mean + variance prediction per call on the fitted C2 model,
    device pointers (no copies), `calls` back-to-back calls.  HBM-bound: 4 N (N+1) + 8 N d + 8 N = 67.4 MB per prediction."""
    torch = rk.torch
    z1 = torch.zeros((1, d), dtype=torch.float64, device=rk.device) + 0.1
    m1 = torch.empty((1, 1), dtype=torch.float64, device=rk.device)
    v1 = torch.empty((1, 1), dtype=torch.float64, device=rk.device)
    for _ in range(5):
        h.predict_mean_var_dev(1, z1.data_ptr(), m1.data_ptr(), v1.data_ptr())
    rk.sync(h)
    h.profile_enable(True, ('vargemm',))
    h.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(calls):
        h.predict_mean_var_dev(1, z1.data_ptr(), m1.data_ptr(), v1.data_ptr())
    rk.sync(h)
    dt = (time.perf_counter() - t0) / calls
    h.profile_enable(False)
    prof = h.profile_read(reset=True)
    k_ms = prof['vargemm'][0] / max(prof['vargemm'][1], 1)
    nbytes = 4.0 * N * (N + 1) + 8.0 * N * d + 8.0 * N
    return {'workload': 'B=1 mean+var predictions on the fitted C2 model (N=%d, d=%d), %d back-to-back calls, device pointers' % (N, d, calls),
            'predictions_per_s': 1.0 / dt, 'us_per_call': dt * 1e6, 'finite': bool(torch.isfinite(m1).all() and torch.isfinite(v1).all()),
            'roofline': {'kernel': 'var_small_kernel<1> (streams the lower triangle of L^-1 once)', 'bound': 'hbm',
                         'achieved': nbytes / (k_ms * 1e-3) * 1e-9 if k_ms > 0 else 0.0, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': nbytes / (k_ms * 1e-3) / 8e12 if k_ms > 0 else 0.0, 'avg_launch_us': k_ms * 1e3,
                         'algorithmic_bytes_per_prediction': nbytes, 'whole_call_GBps': nbytes / dt * 1e-9,
                         'whole_call_frac': nbytes / dt / 8e12, 'traffic': None,
                         'note': 'kernel = HIP events around the variance launch; whole call = crosscov + variance + finish launches back to back'}}


def run_c4(rk, steps, warmup, N=4096, d=6, R=64, iters=4, verify_world1=True):
    """BASELINE config C4: log-marginal likelihood + gradient, R random restarts sharded over the ranks: restart r on
    rank r mod world inside gpmpc_train_multistart, ONE ncclAllGather of the (NLL, theta) table over an RCCL communicator
    the library creates (on a CPU test box: the ranks' tables merged over gloo by gp_mpc_amd.train)."""
    import hashlib
    import numpy as np
    from gp_mpc_amd._lib import Handle
    from gp_mpc_amd.synthetic import synthetic_problem
    from gp_mpc_amd.train import lhs_starts, bounds_ipopt_path, sharded_multistart
    rank, world = rk.rank, rk.world
    p = synthetic_problem(N, d, 1, 1, seed=1234, sn=1e-2)             # the same problem on every rank (replicated X, y)
    h = Handle(rk.lib, p['X'], p['Y'], device=rk.dev_index)
    lb, ub = bounds_ipopt_path(d)
    starts = lhs_starts(R, lb, ub, 1234)[None]
    comm, rccl_ranks = None, 0
    if rk.gpu:
        box = [rk.lib.rccl_unique_id() if rank == 0 else None]
        if rk.dist is not None:
            rk.dist.broadcast_object_list(box, src=0)
        comm = rk.lib.rccl_comm_create(rk.dev_index, world, rank, box[0])     # (world = 1: a self-gather)
        rccl_ranks = rk.lib.rccl_comm_count(comm)
    res = {}

    def step():
        res['r'] = sharded_multistart(h, starts, lb[None], ub[None], max_iter=iters, dist=rk.dist, rank=rank, world=world,
                                      comm=comm, want_invK=False)

    elapsed, _ = timed(rk, h, step, steps, warmup, profile=False)
    if comm is not None:
        rk.lib.rccl_comm_destroy(comm)
    r = res['r']

    def table_hash(rr):
        # the gathered (NLL, theta) table and the arg-min theta*, as bytes: equal hashes = bitwise equal tables
        tab = np.concatenate([np.asarray(rr['obj'], dtype=np.float64)[:, :, None], np.asarray(rr['theta'], dtype=np.float64)], axis=2)
        return (hashlib.sha256(np.ascontiguousarray(tab).tobytes()).hexdigest()[:16],
                hashlib.sha256(np.ascontiguousarray(np.asarray(rr['hyper'], dtype=np.float64)).tobytes()).hexdigest()[:16])

    th, hh = table_hash(r)
    shard_check = {'table_sha16': th, 'theta_star_sha16': hh}
    if world > 1 and verify_world1:
        # the same seeds searched by ONE rank (every restart local, no exchange), on rank 0 after the timed region: the
        # shard is correct iff the two tables are bitwise equal (tests/test_restart_shard.py gates the same on gloo)
        if rank == 0:
            r1 = sharded_multistart(h, starts, lb[None], ub[None], max_iter=iters, dist=None, rank=0, world=1, comm=None,
                                    want_invK=False)
            t1, h1 = table_hash(r1)
            shard_check.update({'world1_table_sha16': t1, 'world1_theta_star_sha16': h1,
                                'bitwise_equal_to_world1': bool(t1 == th and h1 == hh)})
        rk.sync(h)
    out = {
        'metric': 'GP hyper-parameter training restarts/sec, N=%d d=%d fp64 (%d restarts x %d L-BFGS iterations, NLL + analytic gradient)' % (N, d, R, iters),
        'value': R * steps / elapsed, 'unit': 'restarts/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': elapsed / steps * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'C4: %d seeded restarts of the NLL minimisation, restart r on rank r mod world, one all-gather of (NLL, theta)' % R,
                   'N': N, 'd': d, 'restarts': R, 'iterations': iters, 'parallelism': f'restart shard x{world} over RCCL'},
        'rccl_ranks': rccl_ranks, 'shard_check': shard_check,
        'exchange': 'ncclAllGather inside gpmpc_train_multistart' if comm is not None else 'host merge over torch.distributed (no RCCL: test build)',
        'best_nll': float(np.min(r['obj'])), 'finite_restarts': int(np.isfinite(r['obj']).sum()),
        'evaluations_this_rank': int(r['evaluations']), 'restarts_this_rank': len(range(rank, R, world)),
        # algorithmic matrix flops of this rank's share of ONE step (library counter: N^3/3 per Cholesky, per L^-1 formed, per
        # K^-1 lower triangle) against the fp64 MFMA peak over the step's wall time (launch gaps, host optimiser, exchange included)
        'roofline': (lambda gf: {'kernel': 'lock-step batches: two-level Cholesky (+ L^-1, K^-1 for gradient points), NLL / gradient reductions',
                                 'bound': 'mfma', 'achieved': gf * 1e-3 / (elapsed / steps), 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                 'frac': gf * 1e-3 / (elapsed / steps) / FP64_MFMA_PEAK_TFLOPS, 'gflop_per_step_this_rank': gf,
                                 'traffic': None})(float(h.counter('train_gflop'))),
        'device': rk.lib.device_name(rk.dev_index)}
    h.close()
    # self-validation of a multi-GPU record: the communicator that carried the exchange must span every rank, and the
    # sharded search must equal the one-rank search bit for bit -- otherwise the job FAILS (rc != 0, no JSON line)
    if rk.gpu and rccl_ranks != world:
        raise SystemExit('bench.py: the RCCL communicator of the restart shard has %d ranks, the job has %d' % (rccl_ranks, world))
    if shard_check.get('bitwise_equal_to_world1') is False:
        raise SystemExit('bench.py: restart shard over %d ranks differs from the one-rank search of the same seeds: %s' % (world, shard_check))
    return out


def main():
    _own_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--N', type=int, default=4096)
    ap.add_argument('--d', type=int, default=6)
    ap.add_argument('--B', type=int, default=10000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the C3 / C4 steps after the timed C2 region')
    ap.add_argument('--two-calls', action='store_true', help='C2 step as gpmpc_fit + gpmpc_predict_mean_var (r01-r05) instead of ONE gpmpc_fit_predict_mean_var call (same bits; the one call is 0.5-1 % faster: the host round trip between the calls is off the device path)')
    ap.add_argument('--restarts', type=int, default=64, help='restarts of the restart-shard leg (C4)')
    ap.add_argument('--config', default='C2', choices=['C2', 'C3', 'C4'])
    args = ap.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return _launch_ranks(args.gpus)
    import numpy as np
    import torch                     # first: one HIP runtime in the process (torch's), shared by the library
    from gp_mpc_amd._lib import Handle
    from gp_mpc_amd.synthetic import synthetic_problem
    rk = Ranks()
    rank, local_rank, world = rk.rank, rk.dev_index, rk.world
    lib = rk.lib
    if args.config == 'C3':
        out = run_c3(rk, min(args.steps, 10), min(args.warmup, 2), N=args.N if args.N != 4096 else 8192)
    elif args.config == 'C4':
        out = run_c4(rk, min(args.steps, 5), min(args.warmup, 1), N=args.N, d=args.d, R=args.restarts)
    if args.config != 'C2':
        rk.close()
        if rank == 0:
            _emit(out)
        return

    N, d, B = args.N, args.d, args.B
    p = synthetic_problem(N, d, 1, B, seed=1234 + rank, sn=1e-2)
    layout, mfma_rate = lib.mfma_selftest(local_rank)
    h = Handle(lib, p['X'], p['Y'], device=local_rank)
    dev = rk.device
    z = torch.from_numpy(p['Z']).to(dev)
    mean = torch.empty((B, 1), dtype=torch.float64, device=dev)
    var = torch.empty((B, 1), dtype=torch.float64, device=dev)
    hyper = np.ascontiguousarray(p['hyper'])
    h.set_pointer_mode(True)

    two_calls = args.two_calls or os.environ.get('GPMPC_BENCH_TWO_CALLS') == '1'

    def step_two_calls():
        h.fit(hyper)                                                    # K build + Cholesky + L^-1 + alpha
        h.predict_mean_var_dev(B, z.data_ptr(), mean.data_ptr(), var.data_ptr())   # 10k mean+var

    def step_fused():
        # the same work through ONE call of the C ABI (r06, gpmpc_fit_predict_mean_var: same bits as the two calls)
        h.fit_predict_mean_var_dev(hyper, B, z.data_ptr(), mean.data_ptr(), var.data_ptr())

    step = step_two_calls if two_calls else step_fused

    # Timed region: only the dominant kernel is bracketed by events (the roofline needs its launch durations from THIS region);
    # bracketing all seven phases costs 65 us per step (tools/bench_noprof.py), so the phase split comes from a second,
    # untimed pass with all brackets on.
    elapsed, prof_timed = timed(rk, h, step, args.steps, args.warmup, phases=('vargemm',))
    phase_steps = max(5, min(args.steps, 20))
    elapsed_all, prof = timed(rk, h, step, phase_steps, 0)
    prof = {k: (v[0] * args.steps / phase_steps, v[1] * args.steps / phase_steps) for k, v in prof.items()}   # as if over `steps` steps (launch counts scaled, not rounded)
    prof['vargemm'] = prof_timed['vargemm']
    # the other form of the step (two calls / one call), a short untimed-for-the-headline loop, for the record
    elapsed_other, _ = timed(rk, h, step_fused if two_calls else step_two_calls, 10, 2, profile=False)
    step()                                                              # (the outputs the parity legs read: the headline form's)
    rk.sync(h)

    # HBM traffic of the dominant kernel: NOT measured in this run (rocprofv3 --pmc serialises dispatches); it is read
    # from the committed PMC passes under profiles/ and labelled as such
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get('N') == N and tj.get('B') == B:
                traffic = tj['hbm_bytes_per_launch']
                traffic_source = 'profiles/traffic.json (' + str(tj.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, committed')) + ')'
        except Exception:
            traffic = None
    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        gemm_ms, gemm_n = prof['vargemm']
        flops = float(N) * (N + 1) * B              # 2 * N(N+1)/2 MACs per prediction (triangular L^-1)
        achieved = flops / (gemm_ms / max(gemm_n, 1) * 1e-3) * 1e-12 if gemm_ms > 0 else 0.0
        fac_ms, fac_n = prof['factor']
        out = {
            'metric': 'GP predict (mean+var)/sec, N=4096 d=6 fp64', 'value': value, 'unit': 'predictions/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'C2: single-output SE-ARD GP, K build + Cholesky + 10k mean+var predictions per step',
                       'N': N, 'd': d, 'Ny': 1, 'B': B, 'parallelism': f'independent GP per GPU x{world}'},
            'roofline': {'kernel': 'vargemm_persist_kernel: 128x128 tiles over a static schedule, 512 resident workgroups (variance GEMM + column sum of squares)',
                         'bound': 'mfma', 'achieved': achieved, 'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / FP64_MFMA_PEAK_TFLOPS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'avg_launch_ms': gemm_ms / max(gemm_n, 1), 'launches': gemm_n,
                         'peak_measured_mfma_only_ubench': mfma_rate},
            'step_api': 'gpmpc_fit + gpmpc_predict_mean_var' if two_calls else 'gpmpc_fit_predict_mean_var (one call; bitwise the two calls)',
            'ms_per_step_other_api': {'api': 'gpmpc_fit_predict_mean_var' if two_calls else 'gpmpc_fit + gpmpc_predict_mean_var',
                                      'ms_per_step': elapsed_other / 10 * 1e3, 'steps': 10},
            'fused_fit_predicts': int(h.counter('fused_fit_predicts')),
            'phases_ms_per_step': {k: v[0] / args.steps for k, v in prof.items() if v[1] > 0},
            'ms_per_step_all_brackets': elapsed_all / phase_steps * 1e3,   # the second pass: every phase bracketed, as r01-r03's lines were timed
            'runtime': lib.runtime_info(),
            'phases_source': 'second, untimed pass of %d steps with every phase bracketed by events (all brackets on cost ~65 us per step); vargemm and the roofline: from the timed region, the only bracket there' % phase_steps,
            'cholesky': (lambda ms, n: {'kernel': 'chol_chain_kernel (+ chol_worker_kernel x3, concurrent)', 'bound': 'mfma',
                                        'achieved': (N ** 3 / 3.0) / (ms / max(n, 1) * 1e-3) * 1e-12 if ms > 0 else 0.0,
                                        'peak': FP64_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                        'frac': ((N ** 3 / 3.0) / (ms / max(n, 1) * 1e-3) * 1e-12 / FP64_MFMA_PEAK_TFLOPS) if ms > 0 else 0.0,
                                        'avg_launch_ms': ms / max(n, 1), 'launches': n,
                                        'note': 'N^3/3 flops; latency-bound: 64 sequential 64x64 leaves on one CU'})(*prof['chain']),
            'cholesky_plus_inverse': {'ms': fac_ms / max(fac_n, 1),
                                      'tflops': (2.0 * N ** 3 / 3.0) / (fac_ms / max(fac_n, 1) * 1e-3) * 1e-12 if fac_ms > 0 else 0.0,
                                      'note': 'N^3/3 (potrf) + N^3/3 (trtri) flops: blocked right-looking Cholesky + level-batched inverse'},
            'predict_only_per_s': B / max((prof['crosscov'][0] + prof['vargemm'][0] + prof['finish'][0]) / args.steps * 1e-3, 1e-12),
            'device': lib.device_name(local_rank), 'mfma_layout': layout,
            # the persistent chain / worker kernels assume co-residency; a hand-off that times out falls back to the
            # single-queue factorisation (correct, slower): on a shared GPU this count says the fast path degraded
            'handoff_timeouts': int(h.counter('handoff_timeouts')),
            'chained_factorisations': int(h.counter('chained_factorisations')),
            'single_queue_factorisations': int(h.counter('single_queue_factorisations')),
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, cmean, cvar, mscale = cpu_baseline(p, B)
            out['cpu_baseline'] = cb
            gm, gv = mean.cpu().numpy()[:len(cmean), 0], var.cpu().numpy()[:len(cvar), 0]
            out['parity_vs_cpu'] = parity_report(gm, gv, cmean, cvar, mscale, float(p['hyper'][0, d] ** 2))
            out['parity_vs_cpu']['note'] = ('sn = 1e-2: cond(K) ~ 1e7, the CPU path (LU solves) is itself at cond x eps; '
                                            'pointwise-relative figures are dominated by means near zero')
            out['parity_vs_oracle'] = oracle_parity(p, mean.cpu().numpy()[:, 0], var.cpu().numpy()[:, 0], len(cmean))
            # the well-conditioned twin (sn = 0.1) the strict 1e-10 relative bars are gated on (tests: test_c2_full_size_rel_to_max_and_floored_pointwise_bars)
            q = synthetic_problem(N, d, 1, B, seed=1234 + rank, sn=0.1)
            hq = Handle(lib, q['X'], q['Y'], device=local_rank)
            hq.fit(q['hyper'])
            qm, qv = hq.predict_mean_var(q['Z'][:len(cmean)])
            hq.close()
            _, c2m, c2v, ms2 = cpu_baseline(q, B)
            out['parity_vs_cpu_sn0.1'] = parity_report(qm[:, 0], qv[:, 0], c2m, c2v, ms2, float(q['hyper'][0, d] ** 2))
            out['parity_vs_oracle_sn0.1'] = oracle_parity(q, qm[:, 0], qv[:, 0], len(c2m))
    b1 = run_b1(rk, h, N, d) if (world == 1 and not args.no_secondary) else None
    h.close()
    del z, mean, var
    if world > 1:
        # the restart shard, the one part of the path that shards: strong scaling over the ranks, through RCCL
        rs = run_c4(rk, 2, 1, N=N, d=d, R=args.restarts)
        if rank == 0:
            out['restart_shard'] = {k: rs[k] for k in ('metric', 'value', 'unit', 'scaling', 'ms_per_step', 'steps', 'warmup',
                                                        'rccl_ranks', 'shard_check', 'exchange', 'best_nll', 'finite_restarts',
                                                        'evaluations_this_rank', 'restarts_this_rank', 'config')}
    elif not args.no_secondary:
        # driver-witnessed secondary configurations (outside the timed C2 region): 2 steps of C3, 1 of C4
        c3 = run_c3(rk, 2, 1)
        c4 = run_c4(rk, 1, 1, N=N, d=d, R=args.restarts)
        if rank == 0:
            out['secondary'] = {
                'c3': {'workload': c3['config']['workload'], 'ms_per_step': c3['ms_per_step'], 'steps': c3['steps'],
                       'propagation_steps_per_s': c3['value'], 'factor_ms': c3['roofline']['avg_launch_ms'],
                       'factor_tflops': c3['roofline']['achieved'], 'factor_frac': c3['roofline']['frac'],
                       'rollout_ms_per_call': c3['rollout_ms_per_call'], 'finite': c3['finite'],
                       'phases_ms_per_step': c3['phases_ms_per_step'], 'em_pair_kernels': c3['em_pair_kernels'],
                       'rollout_hbm': c3['rollout_hbm']},
                'c5': dict(c3['c5'], car_size=run_c5_small(rk)),
                'b1': b1,
                'c4': {'workload': c4['config']['workload'], 'restarts_per_s': c4['value'], 'ms_per_step': c4['ms_per_step'],
                       'steps': c4['steps'], 'rccl_ranks': c4['rccl_ranks'], 'shard_check': c4['shard_check'], 'exchange': c4['exchange'],
                       'best_nll': c4['best_nll'], 'finite_restarts': c4['finite_restarts'],
                       'evaluations_this_rank': c4.get('evaluations_this_rank'), 'roofline': c4['roofline']}}
    rk.close()
    if rank == 0:
        _emit(out)


if __name__ == '__main__':
    main()
