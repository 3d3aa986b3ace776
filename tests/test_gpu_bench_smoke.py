"""-m gpu: bench.py's contract on small shapes -- one JSON line with the required keys, for one
process and for the torch.distributed.run launch line the driver uses (two ranks sharing the box's
GPU, gloo standing in for RCCL, which refuses two ranks per device)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
        'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'}


def last_json(out):
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_process_line():
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '3', '--warmup', '1', '--ndata', '4096', '--nparam',
                        '256', '--cpu-seconds', '1'], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert KEYS <= set(d)
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['dtype'] == 'f64'
    assert d['value'] > 0 and abs(d['value'] * d['ms_per_step'] / 1e3 - 1) < 1e-9
    rf = d['roofline']
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and 0 < rf['frac'] < 1
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-12
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0
    cm = d['config']['chi2_match']
    assert cm['ok'] and cm['rel_diff'] < 1e-6 and cm['after_lm_steps'] >= 1
    assert d['metric'].endswith('chi2 match vs GSL')
    assert set(cb['phases_s']) == {'trig', 'whiten', 'syrk', 'cholesky', 'other'} and cb['phases_s']['syrk'] > 0
    assert 'other_workloads' not in d['config']            # (debug shapes: the headline's companions are not measured)
    fq = cb['faithful_qr_1thread']
    assert fq['cores'] == 1 and fq['extrapolated'] and 0 < fq['value'] < cb['value'] * 100


def test_two_rank_launch_line():
    env = dict(os.environ, LSQAMD_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--standalone', '--local-addr', '127.0.0.1', 'bench.py', '--gpus', '2', '--steps',
                        '3', '--warmup', '1', '--ndata', '4096', '--nparam', '256'],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert KEYS <= set(d) and d['n_gpus'] == 2 and d['scaling'] == 'strong'
    assert d['phases_calls']['reduce'] >= 2 * d['phases_calls']['jacobian']     # packed normal eqs + trial chi2
    assert d['cpu_baseline'] is None and d['value'] > 0


def test_two_rank_self_launch_through_the_library_collective(tmp_path):
    """`python bench.py --gpus 2` (the self-launch path) with the sums running through comm.hip: the
    RCCL stand-in of tests/fake_rccl.cpp lets both ranks share the one GPU; config.collective must
    name the library's collective and every rank reports its reduce phase."""
    fake = str(tmp_path / 'libfake_rccl.so')
    b = subprocess.run([os.environ.get('HIPCC', 'hipcc'), '-O2', '-std=c++17', '-fPIC', '-shared',
                        os.path.join(ROOT, 'tests', 'fake_rccl.cpp'), '-o', fake, '-lrt'], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    env = dict(os.environ, LSQAMD_DIST_BACKEND='gloo', LSQAMD_COLLECTIVE='rccl', LSQAMD_RCCL_PATH=fake)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--ndata', '8192',
                        '--nparam', '512'], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d['n_gpus'] == 2 and d['config']['collective'].startswith('RCCL reduce-scatter + all-gather inside the library')
    assert len(d['per_rank']['reduce_ms_per_call']) == 2 and all(t > 0 for t in d['per_rank']['reduce_ms_per_call'])
    assert d['phases_calls']['reduce'] >= 2 * d['phases_calls']['jacobian']
    # the exposed share of the exchange is MEASURED (HIP events): one exchange on the step's own stream is waited for entirely
    ex = d['config']['exchange']
    assert len(ex['exposed_share_per_rank']) == 2 and all(abs(e - 1.0) < 1e-9 for e in ex["exposed_share_per_rank"])
    assert d['config']['exchange_exposed_share'] == max(ex['exposed_share_per_rank'])
    # the same run with the exchange in two groups of tile rows on the handle's exchange stream: same bench contract, the
    # share measured again (the stand-in blocks the host inside every collective, so nothing is claimed about its size)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--ndata', '8192',
                        '--nparam', '512', '--whole-fit-maxit', '0'], cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(env, LSQAMD_EXCHANGE_GROUPS='2'))
    assert r.returncode == 0, r.stderr[-2000:]
    g = last_json(r.stdout)
    assert 'in 2 groups' in g['config']['collective'] and g['config']['exchange']['groups'] == 2
    assert all(e is not None and e > 0 for e in g['config']['exchange']['exposed_share_per_rank'])
    assert g['phases_calls']['exch_coll'] == 2 * g['phases_calls']['jacobian'] + (g['phases_calls']['reduce'] - g['phases_calls']['jacobian'])


def test_config5_line():
    """`python bench.py --workload c5`: BASELINE config 5 (128 lockstep fits of 4096 x 512) as a bench line of its own -- step = one
    lockstep round, the batched J^T J launch's roofline from HIP events, chi2 of the first and last fit against the CPU port."""
    r = subprocess.run([sys.executable, 'bench.py', '--workload', 'c5'], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert KEYS <= set(d) and d['n_gpus'] == 1 and d['dtype'] == 'f64' and '(4096,512)' in d['metric']
    c = d['config']
    assert c['all_converged'] and c['rounds_per_sweep'] >= c['nit_min_max'][1] and c['graph_rounds_per_sweep'] >= c['rounds_per_sweep'] - 1
    assert abs(d['ms_per_step'] * d['steps'] / 1e3 - c['ms_per_sweep'] * c['sweeps_timed'] / 1e3) < 1e-9
    rf = d['roofline']
    assert rf['bound'] == 'mfma' and 0.05 < rf['frac'] < 1 and rf['launches'] >= c['nit_min_max'][1]
    assert all(m['ok'] for m in c['chi2_match'])


def test_shard_shape_line_without_companions():
    """`--workload shard8192`: what each of 8 ranks holds of the headline configuration, as a workload name (the companions are
    measured beside the c4 headline only)"""
    r = subprocess.run([sys.executable, 'bench.py', '--workload', 'shard8192', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
                        '--whole-fit-maxit', '0'], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert '(8192,4096)' in d['metric'] and d['cpu_baseline'] is None and 'other_workloads' not in d['config']
    assert 0.5 < d['roofline']['frac'] < 1 and d['roofline']['per_kernel']['J^T J product']['ms'] > 0


def test_self_launch_has_a_wall_clock_limit():
    """A multi-rank run that does not finish is ended (exit 124), never waited for indefinitely."""
    import time
    env = dict(os.environ, LSQAMD_DIST_BACKEND='gloo', LSQAMD_BENCH_TIMEOUT_S='2')
    t0 = time.time()
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--ndata', '4096',
                        '--nparam', '256'], cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert 'LSQAMD_BENCH_TIMEOUT_S' in r.stderr and time.time() - t0 < 120


def test_eight_rank_self_launch_at_the_shard_shape(tmp_path):
    """`python bench.py --gpus 8 --ndata 8192` (P = 4096: what each of 8 GPUs sees of the headline configuration is N = 8192 rows; here 8
    ranks share 8192 rows and ONE GPU through the RCCL stand-in): the launch line, the library collective with 8 ranks, 8 per-rank
    entries.  Correctness of the sharded sums is tests/test_gpu_comm_multi.py's; nothing here measures scaling."""
    fake = str(tmp_path / 'libfake_rccl.so')
    b = subprocess.run([os.environ.get('HIPCC', 'hipcc'), '-O2', '-std=c++17', '-fPIC', '-shared',
                        os.path.join(ROOT, 'tests', 'fake_rccl.cpp'), '-o', fake, '-lrt'], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    env = dict(os.environ, LSQAMD_DIST_BACKEND='gloo', LSQAMD_COLLECTIVE='rccl', LSQAMD_RCCL_PATH=fake)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '8', '--steps', '2', '--warmup', '1', '--ndata', '8192', '--whole-fit-maxit', '0'],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d['n_gpus'] == 8 and d['config']['collective'].startswith('RCCL reduce-scatter + all-gather inside the library')
    assert len(d['per_rank']['reduce_ms_per_call']) == 8 and len(d['per_rank']['ms_per_step']) == 8
    assert all(t > 0 for t in d['per_rank']['reduce_ms_per_call'])
    assert '(65536,4096)' not in d['metric'] and '(8192,4096)' in d['metric']
    assert d['phases_calls']['reduce'] >= 2 * d['phases_calls']['jacobian']
