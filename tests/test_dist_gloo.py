"""CPU, world_size 2, gloo: the batch-shard + final gather path that runs over RCCL on the 8-GPU node."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class FakeGroup:
    """Stands in for the ProcessorGroup: a deterministic function of each row's controls."""

    def __call__(self, feats):
        return feats['amp_0'].sum(dim=(1, 2))[:, None] * torch.arange(1, 9, dtype=torch.float32)[None, :] + \
            feats['reverb_ir'][:, :8]


def _worker(rank, world, port, global_batch, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ddsp_piano_amd import parallel
    g = torch.Generator().manual_seed(0)
    feats = {'amp_0': torch.randn(global_batch, 5, 1, generator=g), 'reverb_ir': torch.randn(global_batch, 16, generator=g)}
    full_ref = FakeGroup()(feats)
    out = parallel.synthesize_sharded(FakeGroup(), feats)
    ok = torch.allclose(out, full_ref) and out.shape == full_ref.shape
    lo, hi = parallel.shard_range(global_batch, world, rank)
    if global_batch % world == 0:
        local = FakeGroup()(parallel.shard_features(feats, world, rank))
        buf = torch.empty(global_batch, 8)
        parallel.gather_audio(local, buf)
        ok = ok and torch.allclose(buf, full_ref) and (hi - lo) == global_batch // world
        buf2 = torch.zeros(global_batch, 8)                    # the overlapped form bench.py uses
        out2, work = parallel.gather_audio(local, buf2, async_op=True)
        work.wait()
        ok = ok and out2 is buf2 and torch.allclose(buf2, full_ref)
    # the final gather to ONE rank (strategy.gather): rank 1 gets the batch, rank 0 only sends
    one = parallel.synthesize_sharded(FakeGroup(), feats, dst=1)
    ok = ok and ((one is None) if rank != 1 else (one.shape == full_ref.shape and torch.allclose(one, full_ref)))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize('global_batch', [8, 7])
def test_shard_and_gather_world2(global_batch):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, global_batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}
