"""A ProcessorGroup call captured once as a HIP graph and replayed (small batches are launch bound).

One call of the polyphonic group is ~25 kernel launches plus three rocFFT executions.  At batch 64 the GPU works for
2 ms and the launches hide behind it; a single 3 s segment (real-time rendering, the streaming blocks of
streaming.py) is 0.2 ms of GPU work behind 0.25-0.3 ms of enqueueing.  Everything on the path is capturable -- no host
synchronisation, no allocation outside torch's graph pool, tables and rocFFT plans cached by shape -- provided the
warm-up runs on the stream the capture uses: rocFFT plans are kept per (shape, stream) (the execution info carries the
stream), and creating one inside a capture invalidates it.

    fast = CapturedGroup(group, example_features, return_outputs_dict=True)
    out = fast(features)            # same keys / shapes / strides as the example; returns the STATIC output tensors

``group`` may be a ProcessorGroup or a NativeGroup (then the graph holds the kernels ddspp_group_run enqueues).  A replay
costs one multi-tensor copy of the inputs into the captured buffers plus the noise draw; a caller that writes its controls
straight into ``fast.inputs`` (the captured buffers, same keys) and calls ``fast()`` pays neither copy.

The outputs are overwritten by the next call: copy what has to outlive it.  FilteredNoise nodes get fresh uniform noise
on every call (drawn outside the graph with the processor's own counter-based generator -- a captured generator call
would replay the same numbers for ever), or the ``noise=`` the caller passes ([B, P, N], as ProcessorGroup).
"""
from __future__ import annotations

import torch

from . import core


def _static_like(tensors):
    """Clones of `tensors` (dict) that keep their storage relationships: tensors that are views of one buffer (the
    per-voice keys of a batched control network, which the batched route reads without stacking) stay views of one."""
    bases, out = {}, {}
    for k, v in tensors.items():
        st = v.untyped_storage()
        key = (st.data_ptr(), v.dtype)
        if key not in bases:
            bases[key] = torch.empty(st.nbytes() // v.element_size(), dtype=v.dtype, device=v.device)
        out[k] = bases[key].as_strided(v.size(), v.stride(), v.storage_offset())
    return out


class CapturedGroup:
    def __init__(self, group, features, return_outputs_dict=False, warmup=2):
        from .synths import FilteredNoise
        feats = {k: core.tf_float32(v) for k, v in features.items()}
        dev = next(iter(feats.values())).device
        if dev.type != 'cuda':
            raise ValueError('CapturedGroup needs device tensors')
        self.group = group
        self.return_outputs_dict = bool(return_outputs_dict)
        self._in = _static_like(feats)
        self._keys = list(self._in)
        torch._foreach_copy_([self._in[k] for k in self._keys], [feats[k] for k in self._keys])
        self._noise_procs = [node[0] for node in group.dag if isinstance(node[0], FilteredNoise)]
        self._stream = torch.cuda.Stream(device=dev)
        self._stream.wait_stream(torch.cuda.current_stream(dev))
        self._noise = None
        # Every rocFFT plan the warm-up and the capture touch is pinned in the library's plan caches for the life of this
        # object: replay() re-executes them without going through the caches, whose LRU policy would otherwise destroy
        # a captured plan (and free its twiddle / work buffers) once enough other shapes have been used.
        from . import effects
        self._plan_caches = (core._plan_cache, effects._irfft_plans)
        recs = [c.record() for c in self._plan_caches]
        self._pinned = [r.__enter__() for r in recs]
        try:
            self._capture(group, dev, warmup)
        finally:
            for r in recs:
                r.__exit__(None, None, None)

    def _capture(self, group, dev, warmup):
        with torch.cuda.stream(self._stream):
            probe = group(self._in, return_outputs_dict=False)                  # shapes; also builds tables and plans
            if self._noise_procs:
                b, n = probe.shape
                self._noise = torch.empty((b, len(self._noise_procs), n), dtype=torch.float32, device=dev)
                self._draw()
            for _ in range(max(int(warmup), 1)):                                # ON the capture stream (see above)
                self._run()
        torch.cuda.current_stream(dev).wait_stream(self._stream)
        torch.cuda.synchronize(dev)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, stream=self._stream):
            self._out = self._run()

    def __del__(self):
        try:
            for cache, entries in zip(getattr(self, '_plan_caches', ()), getattr(self, '_pinned', ())):
                cache.unpin_all(entries)
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def _run(self):
        kw = {'noise': self._noise} if self._noise is not None else {}
        return self.group(self._in, return_outputs_dict=self.return_outputs_dict, **kw)

    @property
    def inputs(self):
        """The captured input buffers ({key: tensor}): written in place, `fast()` replays without copying anything."""
        return self._in

    def _draw(self):
        b, p, n = self._noise.shape
        if (b * p * n) % 4 == 0:                                                  # one counter step per call, as eager
            self._noise_procs[0].draw_noise(b * p, n, self._noise.device, out=self._noise.view(b * p, n))
        else:
            self._noise.copy_(self._noise_procs[0].draw_noise(b * p, n, self._noise.device).view(b, p, n))

    def __call__(self, features=None, noise=None):
        dst, src = [], []
        for k in (self._keys if features is not None else ()):
            v = features[k]
            if v is self._in[k]:                      # the caller wrote into the captured buffer
                continue
            if tuple(v.shape) != tuple(self._in[k].shape):
                raise ValueError(f'{k}: captured for shape {tuple(self._in[k].shape)}, got {tuple(v.shape)}')
            dst.append(self._in[k])
            src.append(v if v.dtype == torch.float32 else core.tf_float32(v))
        if dst:
            torch._foreach_copy_(dst, src)
        if self._noise is not None:
            if noise is None:
                self._draw()
            else:
                self._noise.copy_(core.tf_float32(noise).reshape(self._noise.shape))
        self._graph.replay()
        return self._out
