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
dist.barrier()
torch.cuda.synchronize()
res = {'same': bool(torch.equal(out, ref)), 'gather': bool(torch.equal(buf, ref)), 'uneven': bool(torch.equal(uneven, ref)),
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
    assert res == {'same': True, 'gather': True, 'uneven': True, 'finite': True, 'shape': [3, 40 * 96]}


def test_bench_distributed_branch_runs():
    env = _env(_free_port())
    env['DDSPP_BENCH_DIST'] = '1'
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                        '--batch', '4', '--no-roofline', '--no-cpu-baseline', '--no-single-stream'],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])        # the JSON line must be the LAST line
    assert line['n_gpus'] == 1 and line['value'] > 0 and line['steps'] == 2
