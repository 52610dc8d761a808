"""`python bench.py --gpus N` must start its N ranks itself (the driver calls it without torchrun), print ONE line with
n_gpus = N, and carry the restart shard -- the one part of the path that shards -- in it.  CPU tier: the ranks talk over
gloo and compute on the emulated library (test infrastructure, tests/emu; GPMPC_BENCH_LIB exists for this test only)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..'))


def run_bench(*flags, env=None, timeout=900):
    e = dict(os.environ, GPMPC_BENCH_LIB=os.path.join(HERE, 'emu', '_build', 'libgpmpc_emu.so'))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(flags), capture_output=True, text=True,
                       cwd=ROOT, timeout=timeout, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(1200)
def test_gpus_2_launches_two_ranks_and_reports_the_restart_shard():
    subprocess.check_call([os.path.join(HERE, 'emu', 'build_emu.sh')], stdout=subprocess.DEVNULL)
    j = run_bench('--gpus', '2', '--N', '128', '--B', '64', '--steps', '2', '--warmup', '1', '--restarts', '4',
                  '--no-cpu-baseline')
    assert j['n_gpus'] == 2 and j['steps'] == 2 and j['scaling'] == 'weak'
    assert j['config']['N'] == 128 and j['value'] > 0
    rs = j['restart_shard']
    assert rs['scaling'] == 'strong' and rs['unit'] == 'restarts/s' and rs['value'] > 0
    assert rs['config']['restarts'] == 4 and rs['restarts_this_rank'] == 2      # rank 0 ran restarts 0 and 2
    assert rs['finite_restarts'] == 4                                             # ... and received 1 and 3
    assert rs['rccl_ranks'] == 0 and 'host merge' in rs['exchange']              # no RCCL on a CPU box
    # the record validates itself: the sharded table equals the one-rank search of the same seeds bit for bit
    sc = rs['shard_check']
    assert sc['bitwise_equal_to_world1'] is True and sc['table_sha16'] == sc['world1_table_sha16']
    assert sc['theta_star_sha16'] == sc['world1_theta_star_sha16']


@pytest.mark.timeout(1200)
def test_config_c4_under_an_external_launcher_env():
    """The torch.distributed.run form: the ranks come from the environment, --gpus only names them."""
    subprocess.check_call([os.path.join(HERE, 'emu', 'build_emu.sh')], stdout=subprocess.DEVNULL)
    e = dict(os.environ, GPMPC_BENCH_LIB=os.path.join(HERE, 'emu', '_build', 'libgpmpc_emu.so'), WORLD_SIZE='2',
             MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
    procs = []
    for r in range(2):
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', 'C4', '--N', '96',
                                       '--restarts', '4', '--steps', '1', '--warmup', '0'],
                                      env=dict(e, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1][-2000:] + outs[1][1][-2000:]
    assert outs[1][0].strip() == ''                                              # only rank 0 prints
    j = json.loads(outs[0][0].strip())
    assert j['n_gpus'] == 2 and j['scaling'] == 'strong' and j['finite_restarts'] == 4
