"""CPU tier: run the UNMODIFIED HIP kernel sources under the emulator (tests/emu) and check them
against the oracle at small sizes.  This is test infrastructure for catching kernel-logic bugs
without a GPU; the product never loads this library (gp_mpc_amd/_lib.py::get_lib)."""
import os
import subprocess

import pytest

import parity_cases as pc
from gp_mpc_amd._lib import GpmpcLib

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def emu():
    subprocess.check_call([os.path.join(HERE, 'emu', 'build_emu.sh')], stdout=subprocess.DEVNULL)
    return GpmpcLib(os.path.join(HERE, 'emu', '_build', 'libgpmpc_emu.so'))


def test_emu_mfma_layout(emu):
    layout, _ = emu.mfma_selftest(0)
    assert layout == 0


def test_emu_dgemm(emu):
    pc.check_dgemm(emu)


def test_emu_dgemm_large_tile(emu):
    pc.check_dgemm_large_tile(emu)


def test_emu_forced_tiles(emu, tank):
    pc.check_forced_tiles(emu, tank)


def test_emu_cholesky(emu):
    pc.check_cholesky(emu)


def test_emu_tank_model(emu, tank):
    pc.check_model_fixture(emu, tank, tolL=1e-10, tol_nll=1e-10)


def test_emu_car_model_cond_7e10_bars_5e10_L_1e7_nll(emu, car):
    pc.check_model_fixture(emu, car, tolL=5e-10, tol_nll=1e-7)


def test_emu_gp_class_strict(emu):
    pc.check_gp_class_strict(emu, N=150)


def test_emu_synthetic(emu):
    pc.check_synthetic(emu, N=200, d=6, Ny=2, B=40, sn=0.1, strict_rel=True)
    pc.check_synthetic(emu, N=150, d=3, Ny=1, B=70, sn=1e-2, strict_rel=False)


def test_emu_pipelined_inverse(emu):
    # Np = 576 = 4.5 emulator segments: chained Cholesky with the inverse pipelined behind it (odd tail)
    pc.check_synthetic(emu, N=560, d=4, Ny=2, B=30, sn=0.1, strict_rel=True)


def test_emu_tile_owner_workers(emu):
    # one output: 7 emulated workers own the 45 tiles of Np = 576 (7 each) -> persistent worker kernel +
    # chain kernel + pipelined inverse; Ny = 2 above exceeds the register budget and takes the GEMM path
    pc.check_synthetic(emu, N=560, d=4, Ny=1, B=30, sn=0.1, strict_rel=True)
    pc.check_synthetic(emu, N=300, d=3, Ny=1, B=10, sn=1e-2, strict_rel=False)
    # Np = 704: the workers run as two launches (blocks 0-7 with 7 workers, blocks 8-10 with 2) and the left half
    # of the inverse is computed behind the second one
    pc.check_synthetic(emu, N=700, d=3, Ny=1, B=10, sn=0.1, strict_rel=True)


def test_emu_tile_owner_workers_without_courier(emu):
    # the same schedules with the nine-slot kernel: tile owners make the chain's hand-off tiles themselves
    emu.set_tuning('worker_courier', 0)
    try:
        pc.check_synthetic(emu, N=560, d=4, Ny=1, B=30, sn=0.1, strict_rel=True)
        pc.check_synthetic(emu, N=700, d=3, Ny=1, B=10, sn=0.1, strict_rel=True)
    finally:
        emu.set_tuning('worker_courier', -1)


def test_emu_tile_owner_workers_with_release_fence_handoffs(emu):
    # (the emulator has one coherent memory: this runs the flag protocol of the older publication form, not its fences)
    emu.set_tuning('handoff_write_through', 0)
    try:
        pc.check_synthetic(emu, N=560, d=4, Ny=1, B=30, sn=0.1, strict_rel=True)
    finally:
        emu.set_tuning('handoff_write_through', -1)


def test_emu_three_worker_launches(emu):
    # Np = 960 with 14 emulated workers: blocks 0-7, 8-11 and 12-14 as three worker launches, the inverse of the
    # left half behind the second and of the third quarter behind the third (factor_chain, split3)
    emu.set_tuning('cu_count', 16)
    try:
        pc.check_synthetic(emu, N=950, d=3, Ny=1, B=10, sn=0.1, strict_rel=True)
    finally:
        emu.set_tuning('cu_count', 8)


def test_emu_predict_behind_tail(emu):
    # Np = 768 with 14 emulated workers: two worker launches, so the fit returns at the end of the chain kernel with the
    # last row panel of L^-1 and alpha (workers' queue) in flight; the first prediction forms its cross-covariances on the
    # low-priority queue (throttled grid: 16 workgroups walk the blocks of test points) and its mean next to the
    # variance product; device pointers (host memory under the emulator)
    emu.set_tuning('cu_count', 16)
    try:
        pc.check_predict_behind_tail(emu, N=760, d=3, B=150)
        pc.check_predict_behind_tail(emu, N=650, d=3, B=70, mean_only_second=True, repeats=1)    # (Np = 704: still two worker launches)
    finally:
        emu.set_tuning('cu_count', 8)


def test_emu_variance_persistent_schedule(emu):
    # 16 emulated slots; Np = 320 / 384 -> 3 block rows (the last one partial at 320), 2-3 block columns, two outputs
    pc.check_variance_persistent(emu, N=300, d=3, Ny=2, B=200)
    pc.check_variance_persistent(emu, N=384, d=4, Ny=1, B=330)


def test_emu_jitter_rule(emu, train_small):
    pc.check_jitter_rule(emu, train_small)


def test_emu_nll_gradient(emu, tank):
    pc.check_nll_gradient(emu, tank)


def test_emu_moment_methods(emu, tank):
    pc.check_moment_methods(emu)
    pc.check_moment_methods(emu, tank)


def test_emu_small_batch_chunks(emu):
    pc.check_small_batch_chunks(emu)


def test_emu_timeout_fallback(emu, capfd):
    pc.check_timeout_fallback(emu)
    assert 'timed out on a hand-off' in capfd.readouterr().err


def test_emu_append(emu):
    pc.check_append(emu, N0=300, n=10)      # strip update: rows >= 256 re-factored on top of the stored factors
    pc.check_append(emu, N0=256, n=5, Ny=1)  # old size a multiple of the block: the strip holds new rows only
    pc.check_append(emu, N0=250, n=70)      # too many new rows for the update to pay: refit path
    pc.check_append(emu, N0=40, n=30, Ny=1)  # fewer than 64 old points: refit path


def test_emu_append_after_set_factors_and_rollback(emu):
    pc.check_append_after_set_factors(emu)
    pc.check_append_rollback(emu)


def test_emu_sensitivities(emu, tank, car):
    pc.check_sensitivities(emu, tank)
    pc.check_sensitivities(emu, car, nprobe=5)
    pc.check_sensitivities_batches(emu)


def test_emu_gp_class(emu, tank, tmp_path):
    pc.check_gp_class(emu, tank, tmp_path)


def test_emu_gp_class_car(emu, car, tmp_path):
    from gp_mpc_amd.gp import GP
    import numpy as np
    import gp_oracle as go
    g = car
    gp = GP(g['X'], g['Y'], hyper=dict(hyper=g['hyper'], chol=g['chol'], alpha=g['alpha'], invK=g['invK']),
            normalize=False, gp_method='ME', lib=emu)
    og = go.OracleGP(g['X'], g['Y'], g['hyper'], g['chol'], g['alpha'], g['invK'], gp_method='ME')
    x, u = g['X'][5, :3] * 1.02, g['X'][5, 3:]
    m, c = gp.predict(x, u, np.eye(5) * 1e-6)
    om, oc = og.predict(x, u, np.eye(5) * 1e-6)
    assert np.allclose(m, om, rtol=1e-8, atol=1e-8) and np.allclose(c, oc, rtol=0, atol=1e-10 * (g['hyper'][:, 5] ** 2).max())
    gp.close()


def test_emu_training(emu, train_small):
    pc.check_training(emu, train_small)


def test_emu_edge_cases(emu):
    pc.check_edge_cases(emu)


def test_emu_config_patterns(emu, car):
    """Toy-size runs of the C3 / C4 / C5 bodies the GPU tier executes at full size (tests/test_gpu_configs.py)."""
    import numpy as np
    import gp_oracle as go
    from gp_mpc_amd._lib import Handle
    Ny, d, T = 3, 5, 6
    p = go.synthetic_problem(150, d, Ny, T, seed=12, sn=0.1)
    h = Handle(emu, p['X'], p['Y'])
    assert np.all(h.fit(p['hyper'], want_invK=True) == 0)
    pc.check_rollout_vs_host_loop(h, p, Ny=Ny, d=d, T=T)
    f = h.get_factors()
    pc.check_callback_pattern(h, p['X'], p['hyper'], f['alpha'], f['chol'], p['Z'][:6], p['Sigma'][:6], repeats=1)
    h.close()
    pc.check_rollout_vs_oracle(emu, N=120, Ny=3, d=5, T=8)
    q = go.synthetic_problem(90, 3, 1, 1, seed=4, sn=1e-2)
    pc.check_random_restarts(emu, q['X'], q['Y'], multistart=4, maxiter=3, min_finite=2)
    # (check_two_handles_two_threads is GPU-only: the fiber emulator keeps its scheduler state in globals)


def test_emu_mean_functions(emu):
    pc.check_mean_functions(emu)


def test_emu_feedback_rollout(emu, tank):
    pc.check_feedback_rollout(emu, tank)


def test_emu_em_sens(emu):
    pc.check_em_sens(emu, N=100, d=3, Ny=2, B=2)
    pc.check_em_sens(emu, N=70, d=8, Ny=1, B=1, seed=3)
    pc.check_em_sens(emu, N=100, d=1, Ny=3, B=2, seed=7)      # one input dimension
    pc.check_em_sens(emu, N=47, d=7, Ny=4, B=1, seed=8)       # ragged, nearly the full cross-term depth
    pc.check_em_sens(emu, N=60, d=9, Ny=2, B=1, seed=9)       # d > 8: the 16-deep instantiation of the kernels
    pc.check_em_sens(emu, N=40, d=16, Ny=1, B=1, seed=10)


def test_emu_callback_blocks(emu):
    pc.check_callback_blocks(emu, N=60, Ny=2, Nu=1)


def test_emu_callback_all_nodes_in_one_call(emu):
    pc.check_callback_batched(emu, N=50, Ny=2, Nu=1, Nt=3)


@pytest.mark.parametrize('version', ['3.4.5', '3.6.3'])
def test_emu_callback_classes_execute_under_stub_casadi(emu, version):
    """casadi_callback.py's four Callback subclasses run through CasADi's Callback protocol (tests/stub_casadi.py), both
    Jacobian conventions; values against OracleGP.predict, Jacobians against differences of the oracle."""
    import stub_casadi
    pc.check_callback_classes(emu, stub_casadi, version, N=50, Ny=2, Nu=1, Nt=2)


def test_callback_layout_follows_the_casadi_version():
    """3.4 / 3.5 (the reference: README.md:18-19) take ONE stacked Jacobian from get_jacobian, >= 3.6 one per pair."""
    from gp_mpc_amd import casadi_callback as cb
    assert cb.casadi_version('3.4.5') == (3, 4) and cb.casadi_version('3.5.5') == (3, 5)
    assert cb.casadi_version('3.6.3+') == (3, 6) and cb.casadi_version('3.7') == (3, 7)
    assert cb.jacobian_layout('3.4.5') == 'dense' and cb.jacobian_layout('3.5.1') == 'dense'
    assert cb.jacobian_layout('3.6.0') == 'blocks' and cb.jacobian_layout('3.7.1') == 'blocks'
    # block-diagonal sparsity of the batched signature, column-major vec layout
    Ny, Nu, Nx, Nt = 2, 1, 3, 3
    sp = cb.batched_block_sparsity(Ny, Nu, Nx, Nt)
    rows, cols, shape = sp[0]                       # d M / d X
    assert shape == (Ny * Nt, Ny * Nt) and len(rows) == Nt * Ny * Ny
    assert set(zip(rows // Ny, cols // Ny)) == {(t, t) for t in range(Nt)}                 # node t only sees node t
    rows, cols, shape = sp[5]                       # d V / d C
    assert shape == (Ny * Ny * Nt, Nx * Nx * Nt) and len(rows) == Nt * Ny * Ny * Nx * Nx
    assert set(zip(rows // (Ny * Ny), cols // (Nx * Nx))) == {(t, t) for t in range(Nt)}
    assert all(len(set(zip(r, c))) == len(r) for r, c, _ in sp)                             # no duplicate entries


def test_emu_training_lockstep_batches_are_composition_independent(emu):
    # (sizes trimmed in r05: the five batch / schedule configurations are the point, not the length of the search)
    pc.check_train_lockstep_invariance(emu, N=300, d=3, nstart=4, max_iter=2)                  # two-level execution (Np = 320)
    pc.check_train_lockstep_invariance(emu, N=100, d=2, nstart=3, max_iter=2, mean_func='linear')   # flagged-GEMM execution, trained mean


def test_emu_training_native(emu, train_small):
    pc.check_training_native(emu, train_small)


def test_emu_random_shapes(emu):
    pc.check_random_shapes(emu, n_cases=8, nmax=160)


def test_emu_rollout_replay(emu):
    pc.check_rollout_replay(emu)


def test_emu_io_pack_boundary(emu):
    pc.check_io_pack_boundary(emu)


def test_emu_wide_inputs(emu):
    pc.check_wide_inputs(emu)


def test_emu_training_active_bound(emu, train_small2):
    pc.check_training_active_bound(emu, train_small2)


def test_emu_training_beats_failed_reference_search(emu, train_small3):
    pc.check_training_beats_failed_reference_search(emu, train_small3)


def test_emu_training_never_worse(emu):
    pc.check_training_never_worse(emu, n_cases=4)


def test_emu_set_factors_then_persistent_mean(emu):
    pc.check_set_factors_persistent_mean(emu)


def test_emu_old_me_reference_pin(emu, tank, car, old_me_pins):
    pc.check_old_me_reference_pin(emu, tank, old_me_pins['tank'])
    pc.check_old_me_reference_pin(emu, car, old_me_pins['car'])


def test_emu_em_pair_sum_chunks(emu):
    pc.check_em_chunks(emu)


@pytest.mark.parametrize('name', ['tank', 'car'])
def test_emu_ta_reference_run_pin(emu, name, request, ta_pins):
    pc.check_ta_reference_pin(emu, request.getfixturevalue(name), ta_pins[name], name)


@pytest.mark.parametrize('name', ['train_small', 'em_model2'])
def test_emu_em_reference_run_pin(emu, name, em_pins):
    pc.check_em_reference_pin(emu, *em_pins[name], name)


def test_emu_reference_written_model_file(emu, ref_written, tmp_path):
    pc.check_reference_written_model(emu, *ref_written, tmp_path)


def test_emu_fused_fit_predict(emu):
    """gpmpc_fit_predict_mean_var: worker path with the cross-covariances in the last launch's window (Np = 704: two launches),
    a single launch (no window: at the chain's end), a two-output model (GEMM path, no early status), the jitter retry."""
    pc.check_fused_fit_predict(emu, N=700, d=3, B=70, repeats=1)
    pc.check_fused_fit_predict(emu, N=300, d=3, B=80, Ny=2, repeats=1)
    pc.check_fused_fit_predict(emu, N=560, d=4, B=70, jitter_case=True, repeats=1)
    pc.check_fused_fit_predict(emu, N=150, d=3, B=40, expect_fused=False, repeats=1)      # B <= 64: the two calls


def test_emu_rollout_multi(emu):
    pc.check_rollout_multi(emu)


def test_emu_mean_and_em_against_extended_precision(emu):
    pc.check_mean_against_extended_precision(emu, N=300, d=4, B=120, sn=1e-2, nprobe=60)
    pc.check_em_against_extended_precision(emu, N=150, d=3, Ny=2, nodes=(1,))


def test_emu_far_points_and_zero_signal_variance(emu):
    pc.check_far_points(emu)


def test_emu_gp_rollout_lockstep_route(emu):
    pc.check_gp_rollout_lockstep(emu)
    pc.check_gp_rollout_lockstep(emu, normalize=False, N=150)
