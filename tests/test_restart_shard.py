"""CPU tier, multi-process: the hyper-parameter restart shard with world_size 2 over
gloo must give bit-identical (theta*, NLL*, factors) to the single-process run on the same seeded
restart list, and must split the optimiser work between the ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def run(world, out_dir, multistart, port, optimizer='scipy'):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = []
    for r in range(world):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker.py'), str(out_dir), str(multistart), optimizer],
                                      env=e))
    for p in procs:
        assert p.wait(timeout=600) == 0


@pytest.mark.timeout(900)
def test_two_rank_shard_matches_single_process(tmp_path):
    subprocess.check_call([os.path.join(HERE, 'emu', 'build_emu.sh')], stdout=subprocess.DEVNULL)
    run(1, tmp_path, 4, 29511)
    run(2, tmp_path, 4, 29512)
    one = np.load(tmp_path / 'scipy_rank0_of1.npz')
    r0 = np.load(tmp_path / 'scipy_rank0_of2.npz')
    r1 = np.load(tmp_path / 'scipy_rank1_of2.npz')
    for k in ('hyper', 'obj', 'chol', 'alpha'):
        assert np.array_equal(one[k], r0[k]), k            # bitwise: same restarts, same arithmetic
        assert np.array_equal(r0[k], r1[k]), k             # every rank ends with the same model
    assert np.all(np.isfinite(one['obj']))
    assert r0['n_eval'] < one['n_eval'] and r1['n_eval'] < one['n_eval']     # the work was sharded
    assert abs(int(r0['n_eval']) + int(r1['n_eval']) - int(one['n_eval'])) == 0


@pytest.mark.timeout(900)
def test_two_rank_shard_native_optimizer(tmp_path):
    """The same with `gpmpc_train_multistart` (restart r on rank r mod world inside the C call; no RCCL on a CPU box,
    so the ranks' rows are merged over gloo and the fit is issued afterwards)."""
    subprocess.check_call([os.path.join(HERE, 'emu', 'build_emu.sh')], stdout=subprocess.DEVNULL)
    run(1, tmp_path, 4, 29513, 'native')
    run(2, tmp_path, 4, 29514, 'native')
    one = np.load(tmp_path / 'native_rank0_of1.npz')
    r0 = np.load(tmp_path / 'native_rank0_of2.npz')
    r1 = np.load(tmp_path / 'native_rank1_of2.npz')
    for k in ('hyper', 'obj', 'chol', 'alpha'):
        assert np.array_equal(one[k], r0[k]), k
        assert np.array_equal(r0[k], r1[k]), k
    assert np.all(np.isfinite(one['obj']))


@pytest.mark.timeout(900)
@pytest.mark.parametrize('optimizer', ['native', 'scipy'])
def test_one_rank_failure_reaches_every_rank(tmp_path, optimizer):
    """A device failure on ONE rank (injected: its second NLL evaluation returns GPMPC_EHIP) must not strand the other in
    the exchange: the failing rank still joins it, with +inf rows and its error code, and BOTH raise."""
    subprocess.check_call([os.path.join(HERE, 'emu', 'build_emu.sh')], stdout=subprocess.DEVNULL)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29521 if optimizer == 'native' else 29522), WORLD_SIZE='2',
               GPMPC_TESTING='1')          # (arms the fault-injection knob: refused in a process without it)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker_fail.py'), str(tmp_path), optimizer],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0          # (a dead-lock shows up as this time-out)
    v0 = open(tmp_path / f'fail_{optimizer}_rank0.txt').read()
    v1 = open(tmp_path / f'fail_{optimizer}_rank1.txt').read()
    assert v1.startswith('GpmpcError -2') and 'injected device failure' in v1, v1
    assert v0.startswith('GpmpcError -2') and 'rank 1' in v0, v0
