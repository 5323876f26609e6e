"""Multi-GPU execution of the synthesis path: shard the batch, gather the audio.

Segments (batch rows) are independent units, so the 8 x MI355X node is used data-parallel: one process
per GPU (torch.distributed, backend "nccl" = RCCL over xGMI), rank r synthesises the contiguous block
of rows shard_range(B, world, r), and the only collective is the final gather of the [B/G, N]
float32 audio (18.4 MB per GPU at batch 64 x 3 s) -- to one rank (dst) or to all of them.  This mirrors what the reference does with
tf.distribute.MirroredStrategy + ``strategy.gather(outputs, axis=0)``
(train_single_phase.py:88-102, evaluate_model.py:32-46) without any gradient traffic.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(global_batch, world_size, rank):
    """Contiguous block of rows owned by ``rank`` (blocks differ by at most one row)."""
    if not 0 <= rank < world_size:
        raise ValueError(f'rank {rank} outside world of size {world_size}')
    base, rem = divmod(int(global_batch), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_features(features, world_size, rank, batch_axis=0, global_batch=None):
    """Slice every batched tensor of a ProcessorGroup feature dict to this rank's rows (views, no copies).

    The batch size is ``global_batch`` or, by default, the leading size of the 3-D control tensors ([B, T, C]).
    Tensors whose leading size is something else pass through whole: a reverb impulse response shared by all rows
    ([L], or [1, L] -- ddsp.effects.Reverb accepts both), scalars, non-tensors."""
    tensors = [v for v in features.values() if isinstance(v, torch.Tensor) and v.dim() > 0]
    if global_batch is None:
        sizes = {v.shape[batch_axis] for v in tensors if v.dim() >= 3}
        if len(sizes) != 1:
            raise ValueError(f'features disagree on the batch size: {sorted(sizes)}')
        global_batch = sizes.pop()
    lo, hi = shard_range(global_batch, world_size, rank)
    out = {}
    for k, v in features.items():
        batched = isinstance(v, torch.Tensor) and v.dim() > 1 and v.shape[batch_axis] == global_batch
        out[k] = v.narrow(batch_axis, lo, hi - lo) if batched else v
    return out


class _Done:
    """A finished collective (the host-staged gloo path is synchronous)."""

    def wait(self):
        return True


def gather_audio(local_audio, out=None, group=None, async_op=False, dst=None):
    """Gather equally sized [B_local, N] audio blocks into [world * B_local, N] (rank order).

    dst=None: an all-gather, every rank gets the whole batch.  dst=r: the reference's ``strategy.gather(outputs,
    axis=0)`` (evaluate_model.py:45: ONE program holds the result) -- only rank r receives (7 x 18 MB over its seven
    xGMI links at batch 64 x 3 s, every other rank just sends its block) and the other ranks get None.

    async_op=True returns (out, work): the collective runs on RCCL's own stream, so the next segment's kernels
    overlap with it; call work.wait() (a stream-level wait) before reading ``out``."""
    world = dist.get_world_size(group)
    local_audio = local_audio.contiguous()
    receives = dst is None or dist.get_rank(group) == dst
    if dst is not None and not 0 <= dst < world:
        raise ValueError(f'dst={dst} outside world of size {world}')
    if out is None and receives:
        out = torch.empty((world * local_audio.shape[0],) + tuple(local_audio.shape[1:]),
                          dtype=local_audio.dtype, device=local_audio.device)
    if not receives:
        out = None
    if local_audio.is_cuda and dist.get_backend(group) == 'gloo':
        # gloo has no device collectives: stage through the host.  Only met when several ranks share ONE GPU (RCCL refuses
        # that), i.e. the multi-rank tests and bench.py's dry run on a single-GPU box; on the node the backend is RCCL.
        if dst is None:
            host = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather(list(host.chunk(world, dim=0)), local_audio.cpu(), group=group)
        else:
            host = torch.empty(out.shape, dtype=out.dtype) if receives else None
            dist.gather(local_audio.cpu(), list(host.chunk(world, dim=0)) if receives else None, dst=_global_rank(dst, group),
                        group=group)
        if receives:
            out.copy_(host)
        return (out, _Done()) if async_op else out
    if dst is not None:
        try:
            work = dist.gather(local_audio, list(out.chunk(world, dim=0)) if receives else None, dst=_global_rank(dst, group),
                               group=group, async_op=async_op)
            return (out, work) if async_op else out
        except NotImplementedError:        # a backend without gather (raised at dispatch, on every rank alike): all-gather,
            full = out if receives else torch.empty((world * local_audio.shape[0],) + tuple(local_audio.shape[1:]),
                                                    dtype=local_audio.dtype, device=local_audio.device)   # keep rank dst's copy
            work = dist.all_gather_into_tensor(full, local_audio, group=group, async_op=async_op)
            return (out, work) if async_op else out
    try:
        work = dist.all_gather_into_tensor(out, local_audio, group=group, async_op=async_op)
    except (RuntimeError, NotImplementedError):
        chunks = list(out.chunk(world, dim=0))
        work = dist.all_gather(chunks, local_audio, group=group, async_op=async_op)
    return (out, work) if async_op else out


def _global_rank(rank_in_group, group):
    return rank_in_group if group is None else dist.get_global_rank(group, rank_in_group)


def gather_audio_uneven(local_audio, global_batch, group=None, dst=None):
    """Gather when global_batch % world != 0: pad to the largest shard, gather, drop the padding."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    biggest = max(shard_range(global_batch, world, r)[1] - shard_range(global_batch, world, r)[0]
                  for r in range(world))
    pad = biggest - local_audio.shape[0]
    x = torch.nn.functional.pad(local_audio, (0, 0, 0, pad)) if pad else local_audio
    full = gather_audio(x, group=group, dst=dst)
    if full is None:
        return None
    rows = []
    for r in range(world):
        lo, hi = shard_range(global_batch, world, r)
        rows.append(full[r * biggest: r * biggest + (hi - lo)])
    return torch.cat(rows, dim=0)


def synthesize_sharded(processor_group, features, group=None, dst=None):
    """features hold the GLOBAL batch on every rank; returns the global audio on every rank (dst=None) or on rank
    ``dst`` only (None elsewhere)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = {v.shape[0] for v in features.values() if isinstance(v, torch.Tensor) and v.dim() >= 3}
    if len(sizes) != 1:
        raise ValueError(f'features disagree on the batch size: {sorted(sizes)}')
    global_batch = sizes.pop()
    local = processor_group(shard_features(features, world, rank, global_batch=global_batch))
    if global_batch % world == 0:
        return gather_audio(local, group=group, dst=dst)
    return gather_audio_uneven(local, global_batch, group=group, dst=dst)


# ---- one long file over several GPUs: shard TIME (ddsp_piano synthesize_midi_file.py renders one file per call) -------
def time_shard_range(n_frames, world_size, rank, block=125):
    """Frames [lo, hi) of a file of n_frames control frames owned by ``rank``: contiguous, whole blocks of `block`
    frames (1000-sample chunk boundaries, streaming.block_frames), the remainder on the last rank."""
    if not 0 <= rank < world_size:
        raise ValueError(f'rank {rank} outside world of size {world_size}')
    n_blocks = n_frames // block
    lo_b, hi_b = shard_range(n_blocks, world_size, rank)
    lo, hi = lo_b * block, hi_b * block
    if rank == world_size - 1:
        hi = n_frames
    return lo, hi


def synthesize_time_sharded(make_synth, features, group=None, noise=None):
    """Every rank holds the file's controls ({key_i: [B, T, C]}); rank r renders its time range from scratch
    (streaming.render_range: a phase-only pass gives the oscillator state at its first frame, the reverb history is
    rendered L samples early) and the pieces are all-gathered along time.  Returns [B, T * U] on every rank."""
    from . import streaming
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    syn = make_synth()
    t = next(v.shape[1] for k, v in features.items() if k not in syn.rkeys)
    lo, hi = time_shard_range(t, world, rank, syn.block)
    mine = streaming.render_range(make_synth, features, lo, hi, noise=noise) if hi > lo else None
    b = next(v.shape[0] for k, v in features.items() if k not in syn.rkeys)
    sizes = [(time_shard_range(t, world, r, syn.block)[1] - time_shard_range(t, world, r, syn.block)[0]) * syn.U
             for r in range(world)]
    biggest = max(sizes)
    dev = next(v.device for k, v in features.items() if k not in syn.rkeys)
    pad = torch.zeros((b, biggest), dtype=torch.float32, device=dev)
    if mine is not None:
        pad[:, :mine.shape[1]] = mine
    full = gather_audio(pad.t().contiguous(), group=group)          # [world * biggest, B]: rank-major along time
    parts = [full[r * biggest: r * biggest + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0).t().contiguous()
