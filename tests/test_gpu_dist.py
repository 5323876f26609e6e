"""GPU, RCCL: the batch-shard + all-gather path on the one GPU a test box has (a single-rank "nccl" group).

The world_size-2 logic is covered on CPU by tests/test_dist_gloo.py; this checks that the same calls are
accepted by the RCCL backend on device buffers produced by the HIP path (and that bench.py's distributed
branch runs), which the 8-GPU scaling run depends on.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from util import synth_controls, synth_ir

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


_WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ['DDSPP_ROOT'], 'tests')); sys.path.insert(0, os.environ['DDSPP_ROOT'])
from util import synth_controls, synth_ir
import ddsp_piano_amd as dp
from ddsp_piano_amd import parallel
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
rng = np.random.default_rng(5)
B, P, T, H, K, L, sr = 3, 4, 40, 96, 64, 3000, 24000
feats = {}
for i in range(P):
    for k, v in synth_controls(rng, B, T, H, S=1, K=K).items():
        feats[f'{k}_{i}'] = torch.as_tensor(v, device='cuda')
feats['reverb_ir'] = torch.as_tensor(synth_ir(rng, B, L), device='cuda')
def group():
    a = dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True)
    n = dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr, seed=3)
    r = dp.Reverb(name='reverb')
    return dp.ProcessorGroup(dp.polyphonic_dag(a, n, r,
        additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
        noise_controls=['magnitudes'], reverb_controls=['reverb_ir'], n_synths=P))
ref = group()(feats)
out = parallel.synthesize_sharded(group(), feats)
buf = torch.empty_like(ref)
parallel.gather_audio(ref, buf)
uneven = parallel.gather_audio_uneven(ref, B)
to_one, work = parallel.gather_audio(ref, dst=0, async_op=True)      # dist.gather over RCCL
work.wait()
# bench.py's `pipelined` loop: the gather of step i is issued async_op=True into one of TWO landing buffers and only
# waited for (a stream-level wait) before the gather of step i + 1, so it runs on RCCL's stream under the next step's
# kernels; every landed buffer must hold ITS step's audio
pg, bufs, state, landed, wants = group(), [torch.empty_like(ref) for _ in range(2)], {'work': None}, [], []
for i in range(10):
    f = dict(feats)
    f['amplitudes_0'] = feats['amplitudes_0'] + 0.05 * i          # a different render every step
    pg.noise.seed = 100 + i
    audio = pg(f)
    wants.append(audio.clone())
    if state['work'] is not None:
        state['work'].wait()
        landed.append(bufs[(i - 1) & 1].clone())
    _, state['work'] = parallel.gather_audio(audio, bufs[i & 1], async_op=True, dst=0)
state['work'].wait()
landed.append(bufs[9 & 1].clone())
pipelined = all(torch.equal(a, b) for a, b in zip(landed, wants)) and len(landed) == 10 and \
    not torch.equal(wants[0], wants[1])
dist.barrier()
torch.cuda.synchronize()
res = {'same': bool(torch.equal(out, ref)), 'gather': bool(torch.equal(buf, ref)) and bool(torch.equal(to_one, ref)),
       'uneven': bool(torch.equal(uneven, ref)), 'pipelined': bool(pipelined),
       'finite': bool(torch.isfinite(out).all()), 'shape': list(out.shape)}
dist.destroy_process_group()
print('RESULT ' + json.dumps(res))
'''


def _env(port):
    env = dict(os.environ)
    env.update({'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'HSA_ENABLE_IPC_MODE_LEGACY': '0',
                'DDSPP_ROOT': ROOT, 'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'})
    return env


def test_single_rank_rccl_shard_and_gather():
    p = subprocess.run([sys.executable, '-c', _WORKER], env=_env(_free_port()), capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith('RESULT ')][-1]
    res = json.loads(line[len('RESULT '):])
    assert res == {'same': True, 'gather': True, 'uneven': True, 'pipelined': True, 'finite': True, 'shape': [3, 40 * 96]}


def test_bench_distributed_branch_runs():
    env = _env(_free_port())
    env['DDSPP_BENCH_DIST'] = '1'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                        '--batch', '4', '--no-roofline', '--no-cpu-baseline', '--no-single-stream', '--sustain-seconds', '0.3'],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])        # the JSON line must be the LAST line
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['steps'] == 2
    # round 5: what the first multi-GPU run reads out by itself -- every rank's step, the collective alone, is it hidden?
    assert len(line['per_rank']['ms_step_with_sync_gather']) == 1 and line['per_rank']['ms_step_compute_only'][0] > 0
    assert set(line['gather_hidden']) >= {'compute_only_ms', 'with_synchronous_gather_ms', 'pipelined_ms', 'hidden'}
    assert line['sustained']['n'] >= 200 and line['sustained']['ms_per_step'] > 0 and 'gpu' in line['sustained']


_WORKER2 = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ['DDSPP_ROOT'], 'tests')); sys.path.insert(0, os.environ['DDSPP_ROOT'])
from util import synth_controls, synth_ir
import ddsp_piano_amd as dp
from ddsp_piano_amd import parallel, streaming
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
sr, P, H, K, L = 24000, 3, 64, 96, 3000
KEYS = dict(additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'], noise_controls=['magnitudes'],
            reverb_controls=['reverb_ir'])
def procs():
    return (dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True),
            dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr, seed=3), dp.Reverb(name='reverb'))
def feats_of(B, T, seed):
    rng = np.random.default_rng(seed)                       # the same on every rank
    f = {}
    for i in range(P):
        for k, v in synth_controls(rng, B, T, H, S=1, K=K, silent_frac=0.0).items():
            f[f'{k}_{i}'] = torch.as_tensor(v, device='cuda')
    f['reverb_ir'] = torch.as_tensor(synth_ir(rng, B, L), device='cuda')
    noise = torch.as_tensor(rng.uniform(-1, 1, [B, P, T * 96]).astype(np.float32), device='cuda')
    return f, noise
res = {}
for name, B in (('even', 4), ('uneven', 3)):
    feats, noise = feats_of(B, 40, 5 + B)
    pg = dp.ProcessorGroup(dp.polyphonic_dag(*procs(), n_synths=P, **KEYS))
    ref = pg(feats, noise=noise)
    class Local:                                            # the shard's rows of the explicit noise go with its controls
        def __call__(self, f):
            lo, hi = parallel.shard_range(B, world, rank)
            return dp.ProcessorGroup(dp.polyphonic_dag(*procs(), n_synths=P, **KEYS))(f, noise=noise[lo:hi])
    out = parallel.synthesize_sharded(Local(), feats)
    res[name] = [list(out.shape) == list(ref.shape), float((out - ref).abs().max() / ref.abs().max())]
    one = parallel.synthesize_sharded(Local(), feats, dst=world - 1)          # the final gather to one rank
    res[name][0] = res[name][0] and ((one is None) if rank != world - 1 else bool(torch.equal(one, out)))
# one file, time sharded: 5 blocks of 125 frames -> 3 + 2 (the 25 remainder frames go to the last rank)
feats, noise = feats_of(1, 650, 9)
ref = dp.ProcessorGroup(dp.polyphonic_dag(*procs(), n_synths=P, **KEYS))(feats, noise=noise)
make = lambda: streaming.StreamingSynthesizer(*procs(), n_synths=P)
out = parallel.synthesize_time_sharded(make, feats, noise=noise)
res['time'] = [list(out.shape) == list(ref.shape), float((out - ref).abs().max() / ref.abs().max())]
res['ranges'] = [list(parallel.time_shard_range(650, world, r, 125)) for r in range(world)]
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print('RESULT ' + json.dumps(res))
"""


def test_two_ranks_on_one_gpu_shard_batch_and_time():
    """multi-rank evidence that runs on a 1-GPU box.  Two processes share cuda:0 (gloo: RCCL refuses
    two ranks on one device; the gather is staged through the host) and drive the REAL ProcessorGroup through
    parallel.synthesize_sharded -- even and uneven batch -- and parallel.synthesize_time_sharded; every rank must hold
    the unsharded render."""
    port = _free_port()
    procs = []
    for rank in range(2):
        env = _env(port)
        env.update({'RANK': str(rank), 'WORLD_SIZE': '2', 'LOCAL_RANK': str(rank)})
        procs.append(subprocess.Popen([sys.executable, '-c', _WORKER2], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    for so, _ in outs:
        res = json.loads([l for l in so.splitlines() if l.startswith('RESULT ')][-1][len('RESULT '):])
        assert res["ranges"] == [[0, 375], [375, 650]]
        for k in ('even', 'uneven'):
            assert res[k][0] and res[k][1] < 1e-6, (k, res[k])            # rows do not depend on the batch they are in
        assert res['time'][0] and res['time'][1] < 3e-5, res['time']     # FFT sizes / summation order differ, nothing else


def test_bench_two_rank_launcher_flow_on_one_gpu():
    """`python bench.py --gpus 2` end to end: the script spawns its two ranks (torch.distributed.run), they shard, gather and
    rank 0 prints ONE JSON line.  On this one-GPU box the ranks share the device and the backend is gloo -- the line says
    so; with 2+ GPUs and no overrides the same flow runs over RCCL."""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update({'HSA_ENABLE_IPC_MODE_LEGACY': '0', 'DDSPP_BENCH_SHARE_GPU': '1', 'DDSPP_BENCH_BACKEND': 'gloo'})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                        '--batch', '4', '--no-roofline', '--no-cpu-baseline', '--no-extras', '--sustain-seconds', '0.2'],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['shared_gpu'] is True and line['backend'] == 'gloo'
    assert line['config']['global_batch'] == 8 and line['value'] > 0 and line['steps'] == 3
    assert line['allgather']['bytes_received'] == 4 * 72000 * 4 and line['gather_to_rank0']['ms'] > 0
    assert line['gather'].startswith('to rank 0')
    assert len(line['per_rank']['ms_step_pipelined']) == 2 and isinstance(line['gather_hidden']['hidden'], bool)
    assert line['sustained']['ms_per_step'] > 0                  # (the sustained loop runs the pipelined step, gathers included)


_WORKER8 = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ['DDSPP_ROOT'], 'tests')); sys.path.insert(0, os.environ['DDSPP_ROOT'])
from util import synth_controls, synth_ir
import ddsp_piano_amd as dp
from ddsp_piano_amd import parallel
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
sr, P, H, K, L, B, T = 24000, 2, 32, 96, 2000, 13, 30
KEYS = dict(additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'], noise_controls=['magnitudes'],
            reverb_controls=['reverb_ir'])
rng = np.random.default_rng(13)                             # the same on every rank
feats = {}
for i in range(P):
    for k, v in synth_controls(rng, B, T, H, S=1, K=K, silent_frac=0.0).items():
        feats[f'{k}_{i}'] = torch.as_tensor(v, device='cuda')
feats['reverb_ir'] = torch.as_tensor(synth_ir(rng, B, L), device='cuda')
noise = torch.as_tensor(rng.uniform(-1, 1, [B, P, T * 96]).astype(np.float32), device='cuda')
def group():
    return dp.ProcessorGroup(dp.polyphonic_dag(
        dp.MultiInharmonic(name='additive', frame_rate=250, sample_rate=sr, inference=True),
        dp.DynamicSizeFilteredNoise(name='noise', frame_rate=250, sample_rate=sr), dp.Reverb(name='reverb'), n_synths=P, **KEYS))
lo, hi = parallel.shard_range(B, world, rank)
class Local:
    def __call__(self, f):
        return group()(f, noise=noise[lo:hi])
one = parallel.synthesize_sharded(Local(), feats, dst=0)     # 13 rows over 8 ranks: 2 2 2 2 2 1 1 1, gathered to rank 0
res = {'rows': hi - lo}
if rank == 0:
    ref = group()(feats, noise=noise)
    res['ok'] = [list(one.shape) == list(ref.shape), float((one - ref).abs().max() / ref.abs().max())]
else:
    res['ok'] = [one is None, 0.0]
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print('RESULT ' + json.dumps(res))
"""


def test_eight_ranks_on_one_gpu_uneven_batch_to_rank0():
    """the world-size-8 flow (config 4's rank count) on the one GPU a test box has: eight gloo ranks
    share cuda:0, a global batch of 13 is sharded unevenly (2 2 2 2 2 1 1 1) and gathered to rank 0, which must hold the
    unsharded render; the other ranks get None."""
    port = _free_port()
    procs = []
    for rank in range(8):
        env = _env(port)
        env.update({'RANK': str(rank), 'WORLD_SIZE': '8', 'LOCAL_RANK': str(rank)})
        procs.append(subprocess.Popen([sys.executable, '-c', _WORKER8], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=1200) for p in procs]
    rows = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
        res = json.loads([l for l in so.splitlines() if l.startswith('RESULT ')][-1][len('RESULT '):])
        assert res['ok'][0] and res['ok'][1] < 1e-6, res
        rows.append(res['rows'])
    assert rows == [2, 2, 2, 2, 2, 1, 1, 1]


def test_bench_eight_rank_launcher_flow_on_one_gpu():
    """`python bench.py --gpus 8` end to end on one GPU (ranks share it, gloo): spawn, shard, gather to rank 0, ONE JSON line
    with n_gpus == 8 and the collectives' GB/s next to the xGMI figure they will be read against on the node."""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update({'HSA_ENABLE_IPC_MODE_LEGACY': '0', 'DDSPP_BENCH_SHARE_GPU': '1', 'DDSPP_BENCH_BACKEND': 'gloo'})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1',
                        '--batch', '2', '--seconds', '1', '--no-roofline', '--no-cpu-baseline', '--no-extras', '--sustain-seconds', '0'],
                       env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['shared_gpu'] is True and line['config']['global_batch'] == 16 and line['value'] > 0
    g = line['gather_to_rank0']
    assert g['bytes_received'] == 7 * 2 * 24000 * 4 and g['gb_per_s'] > 0
    assert g['xgmi']['links_used'] == 7 and abs(g['xgmi']['ceiling_gb_per_s'] - 7 * 153) < 1
    assert line['allgather']['xgmi']['frac_of_ceiling'] > 0
    assert len(line['per_rank']['ms_step_compute_only']) == 8 and 'sustained' not in line
