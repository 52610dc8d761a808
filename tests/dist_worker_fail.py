"""Worker for tests/test_restart_shard.py::test_one_rank_failure_reaches_every_rank: rank 1 of 2 is told to fail its second
NLL evaluation with a device error (fault injection knob of the library); BOTH ranks must come back from the training
call with that error instead of one of them waiting in the exchange for ever.  gloo on CPU, emulated library."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)


def main():
    out_dir, optimizer = sys.argv[1], sys.argv[2]
    world = int(os.environ['WORLD_SIZE'])
    rank = int(os.environ['RANK'])
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gp_mpc_amd._lib import GpmpcLib, Handle, GpmpcError, EHIP
    from gp_mpc_amd.train import train_gp
    lib = GpmpcLib(os.path.join(ROOT, 'tests', 'emu', '_build', 'libgpmpc_emu.so'))
    t = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'train_small.npz')))
    h = Handle(lib, t['X'], t['Y'])
    if rank == 1:
        lib.set_tuning('fail_nll_after', 2)
    verdict = 'no error'
    try:
        train_gp(h, t['X'], t['Y'], multistart=4, random_restarts=True, seed=1234, numpy_path_conventions=False,
                 optimizer_opts={'maxiter': 10}, optimizer=optimizer)
    except GpmpcError as e:
        verdict = 'GpmpcError %d: %s' % (e.code, e)
        assert e.code == EHIP
    open(os.path.join(out_dir, f'fail_{optimizer}_rank{rank}.txt'), 'w').write(verdict)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
