"""bench.py's launch contract (no GPU): ``python bench.py --gpus N`` with no WORLD_SIZE in
the environment re-executes itself under torch.distributed.run with N ranks (the driver's
command form), and under an explicit torchrun launch it reads RANK / WORLD_SIZE from the
environment.  The ranks here join a gloo group and measure nothing (ASR_BENCH_CPU_STUB)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, **env):
    e = dict(os.environ, **env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    out = subprocess.run(cmd, env=e, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=240, stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_gpus_flag_self_launches_n_ranks():
    line = _run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '4', '--warmup', '1'],
                ASR_BENCH_CPU_STUB='1')
    assert line == {'stub': True, 'n_gpus': 2, 'steps': 4, 'warmup': 1, 'gpus_arg': 2}


def test_launch_command_is_the_drivers_form():
    line = _run([sys.executable, 'bench.py', '--gpus', '8', '--steps', '5', '--warmup', '2'],
                ASR_BENCH_DRY_LAUNCH='1')
    cmd = line['launch']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '8', '--steps', '5', '--warmup', '2']
    assert os.path.basename(cmd[-7]) == 'bench.py'


def test_explicit_torchrun_form():
    line = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
                 '2', '--master-addr', '127.0.0.1', '--master-port', '29577', 'bench.py', '--gpus',
                 '2', '--steps', '3', '--warmup', '1'], ASR_BENCH_CPU_STUB='1')
    assert line['n_gpus'] == 2 and line['gpus_arg'] == 2 and line['steps'] == 3
