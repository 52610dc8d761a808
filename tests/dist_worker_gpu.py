"""Worker for tests/test_gpu_configs.py::test_c4_restart_shard_rccl_two_gpus: one rank of the restart shard on its
own GPU, torch.distributed backend nccl (= RCCL over xGMI)."""
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
    local = int(os.environ.get('LOCAL_RANK', '0'))
    import torch
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', rank=rank, world_size=world)
    import gp_oracle as go
    from gp_mpc_amd._lib import Handle, get_lib
    from gp_mpc_amd.train import train_gp
    p = go.synthetic_problem(4096, 6, 1, 1, seed=1234, sn=1e-2)
    h = Handle(get_lib(), p['X'], p['Y'], device=local)
    opt = train_gp(h, p['X'], p['Y'], multistart=multistart, random_restarts=True, seed=1234,
                   numpy_path_conventions=False, optimizer_opts={'maxiter': 3}, optimizer=optimizer)
    f = h.get_factors(chol=False)
    np.savez(os.path.join(out_dir, f'gpu_{optimizer}_rank{rank}_of{world}.npz'), hyper=opt['hyper'], obj=opt['obj'],
             alpha=f['alpha'], n_eval=opt['n_eval'])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
