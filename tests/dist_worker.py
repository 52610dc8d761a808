"""Worker for tests/test_restart_shard.py: one rank of the restart shard (gloo on CPU; the compute
library is the HIP emulator build -- test infrastructure, see tests/emu)."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    sys.path.insert(0, p)


def main():
    out_dir, multistart = sys.argv[1], int(sys.argv[2])
    optimizer = sys.argv[3] if len(sys.argv) > 3 else 'scipy'
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from gp_mpc_amd._lib import GpmpcLib, Handle
    from gp_mpc_amd.train import train_gp
    lib = GpmpcLib(os.path.join(ROOT, 'tests', 'emu', '_build', 'libgpmpc_emu.so'))
    t = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'train_small.npz')))
    h = Handle(lib, t['X'], t['Y'])
    opt = train_gp(h, t['X'], t['Y'], multistart=multistart, random_restarts=True, seed=1234,
                   numpy_path_conventions=False, optimizer_opts={'maxiter': 60 if optimizer == 'scipy' else 25},
                   optimizer=optimizer)
    f = h.get_factors()
    np.savez(os.path.join(out_dir, f'{optimizer}_rank{rank}_of{world}.npz'), hyper=opt['hyper'], obj=opt['obj'],
             chol=f['chol'], alpha=f['alpha'], n_eval=opt['n_eval'])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
