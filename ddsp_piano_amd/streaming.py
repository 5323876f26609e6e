"""Piecewise synthesis of one long control sequence with carried state (SURVEY.md 8f-2, 8e note).

``synthesize_midi_file.py`` renders a whole MIDI file -- minutes of controls -- in one call
(ddsp_piano synthesize_midi_file.py:41-73).  The batched group does that too (a 136 s file is one segment), but a caller
that produces controls incrementally, or wants one file's TIME spread over several GPUs, needs what a single call keeps
internally between two samples.  For this path that is exactly:

  * oscillator banks: ddsp.core.angular_cumsum restarts the phase every 1000 samples and adds the float32 running sum
    of the chunks' end phases -- one float per (voice, oscillator), ``core.oscillator_phase_state`` -- provided a piece
    starts on a chunk boundary: pieces are multiples of BLOCK = 1000 / gcd(U, 1000) frames (125 frames = 0.5 s for
    every shipped sample rate); plus ONE frame of look-ahead, because frame t is interpolated towards frame t + 1;
    plus the piece's ABSOLUTE sample position: the reference's bilinear resize forms float32(n) * (T / N) and takes
    its fractional part as the interpolation weight, which rounds differently at n = 12000 and n = 1212000
    (core.walk_weights) -- on a pitch drop of four octaves inside one frame that is 0.03 rad at partial 128.  Past
    core.linear_exact_frames(U) frames (8.7 min at 24 kHz) the product rounds up to the NEXT whole frame for the last
    sample(s) of a frame: those samples carry a mark in the weight table and take row t + 1 itself (round 4; push()
    used to raise there);
  * FilteredNoise: the time-varying FIR reaches Lw - 1 - delay samples back and `delay` samples forward: the piece is
    filtered with ceil((Lw - 1 - delay) / U) frames of context behind it and ceil(delay / U) frames ahead
    (``noise_reach``: one frame each way at 16 / 24 kHz, two at ENSTDkCl's 8 kHz with 64 bands) and cropped.  Noise
    samples are addressed by their absolute position: explicit ``noise=`` rows, or the library's counter-based Philox
    stream, which here is keyed by (row, absolute sample) -- a piece draws the same numbers whichever PUSH renders it
    (it is NOT the stream ProcessorGroup / NativeGroup draw for a one-call render, which is keyed by call);
  * reverb: the last L - 1 samples of the dry mix (overlap-save).

The pieces then equal the one-call render: the oscillator phases bit for bit, the sums and FFTs to float32 round-off
(tests/test_gpu_streaming.py).  ``render_range`` renders any frame range of a file from scratch (phase state by a
phase-only pass over the prefix, reverb history by rendering L samples early): that is the unit of work of a time shard
(parallel.synthesize_time_sharded).
"""
from __future__ import annotations

import math

import torch

from . import _lib, core
from .core import _lib_, _ptr, _stream
from .effects import FeedbackDelayNetwork
from .polyphonic import _stack_voices


def block_frames(upsampling):
    """Frames per 1000-sample-aligned block: pieces handed to the synthesiser are multiples of this."""
    return 1000 // math.gcd(int(upsampling), 1000)


def noise_reach(n_bands, window_size, upsampling):
    """(frames behind, frames ahead) of a piece that the FilteredNoise FIR of a piece's samples touches: the FIR of
    ddsp.core.frequency_filter has Lw = min(window_size, 2 (K - 1)) taps and is advanced by crop_and_compensate_delay's
    `delay`, so output n collects inputs n + delay - (Lw - 1) ... n + delay."""
    ir_size = 2 * (int(n_bands) - 1)
    lw = ir_size if (window_size <= 0 or window_size > ir_size) else int(window_size)
    delay = lw // 2 if core.RECALLED['auto_delay'] == 'half' else (lw - 1) // 2 - 1
    u = int(upsampling)
    return max(0, -(-(lw - 1 - delay) // u)), max(0, -(-delay // u))


class StreamingSynthesizer:
    """additive: MultiInharmonic(inference=True); noise: (DynamicSize)FilteredNoise; reverb: Reverb, a parameter
    holding FeedbackDelayNetwork, FeedbackDelayNetworkApply or None.  Keys as polyphonic_dag (``<control>_<voice>``)."""

    def __init__(self, additive, noise, reverb=None, n_synths=16,
                 additive_controls=('amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'),
                 noise_controls=('magnitudes',), reverb_controls=('reverb_ir',)):
        from .synths import InHarmonic
        if not isinstance(additive, InHarmonic):
            raise ValueError('streaming takes InHarmonic / MultiInharmonic as the additive synthesiser (a SurrogateAdditive '
                             'group renders in one call: ProcessorGroup)')
        if not additive.inference:
            raise ValueError('streaming needs the angular-cumsum oscillator (inference=True): a plain cumsum has no '
                             'bounded state')
        self.additive, self.noise, self.reverb = additive, noise, reverb
        self.P = int(n_synths)
        self.akeys, self.nkeys, self.rkeys = list(additive_controls), list(noise_controls), list(reverb_controls)
        self.U = additive.upsampling
        self.block = block_frames(self.U)
        self.reset()

    def reset(self, frame=0, phase_state=None):
        self.frame = int(frame)                  # absolute index of the next frame to render
        self.phase = phase_state                 # [R, S * H] or None (a signal that starts at frame 0)
        self._buf = None                         # controls not yet rendered, the voices of a control stacked: {name: [P * B, t, C]}
        self._vm = None                          # ... voice major ([P, B]) or segment major ([B, P]) rows: fixed by the first push
        self._noise_buf = None                   # explicit noise not yet used: [B, P, n]
        self._prev = None                        # the frames before `frame` the noise FIR reaches (controls + samples)
        self._tail = None                        # last L - 1 samples of the dry mix
        self._ir = None

    # ------------------------------------------------------------------------------------------ buffering
    def push(self, features, noise=None, final=False):
        """Append control frames ({key_i: [B, t, C]}, the reverb's controls every time or once) and render what can be
        rendered: whole blocks, keeping one frame of look-ahead unless ``final``.  Returns audio [B, n] (n may be 0)."""
        for k in self.rkeys:
            if k in features:
                self._ir = core.tf_float32(features[k])
        # the P voices of a control as ONE buffer of P * B rows (zero-copy when they are slices of one tensor, as the
        # Parallelizer hands them over): a push then costs one concatenation per control, not one per control and voice
        # (a push of 0.5 s spent 1.0 of its 1.4 ms in 160 tiny copies)
        new = {}
        for name in self.akeys + self.nkeys:
            rows, vm = _stack_voices([core.tf_float32(features[f'{name}_{i}']) for i in range(self.P)], self._vm)
            if self._vm is None:
                self._vm = vm
            new[name] = rows
        if self._buf is None:
            self._buf = new
        else:
            self._buf = {k: torch.cat([self._buf[k], new[k]], dim=1) for k in self._buf}
        if noise is not None:
            noise = core.tf_float32(noise)
            self._noise_buf = noise if self._noise_buf is None else torch.cat([self._noise_buf, noise], dim=2)
        first = next(iter(self._buf.values()))
        have = first.shape[1]
        usable = have if final else ((have - self._lookahead()) // self.block) * self.block
        if usable <= 0:
            return torch.empty((first.shape[0] // self.P, 0), dtype=torch.float32, device=first.device)
        return self._render(usable, final)

    def _reach(self):
        k = self._buf[self.nkeys[0]].shape[-1]
        return noise_reach(k, getattr(self.noise, 'window_size', 257), self.U)

    def _lookahead(self):
        """Frames past a piece that must be known to render it: one for the oscillators (frame t is interpolated
        towards frame t + 1), `ahead` for the noise FIR."""
        return max(1, self._reach()[1])

    # ------------------------------------------------------------------------------------------ one piece
    def _rows(self, key, sl, vm=None):
        return self._buf[key][:, sl].contiguous(), self._vm

    def _render(self, nb, final):
        P, U = self.P, self.U
        have = next(iter(self._buf.values())).shape[1]
        back, _ = self._reach()
        look = min(self._lookahead(), have - nb)     # fewer only at the end of the signal: nothing follows there
        # Past core.linear_exact_frames(U) the reference's float32(n) * (T / N) rounds up to the next whole frame for the
        # last sample(s) of a frame; the kernels take those samples from row t + 1 (core.walk_weights marks them at the
        # piece's absolute positions), so a piece still equals the one-call render.  Only hours into a signal does the
        # table stop following the frame walk at all (more than a block of such samples per frame):
        if self.frame + nb + look > core.linear_exact_frames(U) and \
                not core.walkable(nb + look, (nb + look) * U, self.frame * U, (nb + look) * U):
            raise ValueError(f'streaming stops at frame {self.frame}: this far into a signal the reference\'s bilinear '
                             'resize no longer follows the frame walk of the fused kernels (core.walkable); render such '
                             'a file in one call')
        sl = slice(0, nb + look)
        amp, vm = self._rows(self.akeys[0], sl)
        hd, _ = self._rows(self.akeys[1], sl, vm)
        inh, _ = self._rows(self.akeys[2], sl, vm)
        f0, _ = self._rows(self.akeys[3], sl, vm)
        R, Tc, H = hd.shape
        B, S = R // P, f0.shape[-1]
        n = nb * U
        dev = hd.device
        add = self.additive
        # ---- additive: from the carried phase state; the state after this piece for the next one
        ctl = add._controls(amp, hd, inh, f0, want_counts=True, want_shifts=False)
        inh_rows = ctl['_inharm_coef'].reshape(R, Tc)
        mix = core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(R, Tc), ctl['harmonic_distribution'], None,
                                       B, Tc * U, add.sample_rate, voice_major=vm, audible=ctl['_audible'],
                                       inharm_coef=inh_rows, phase_state=self.phase, sample_offset=self.frame * U)[:, :n]
        if not (final and nb == have):
            self.phase = core.oscillator_phase_state(ctl['f0_hz'], n // 1000, U, add.sample_rate, inharm_coef=inh_rows,
                                                     n_harmonics=H, phase_state=self.phase, audible=ctl['_audible'],
                                                     sample_offset=self.frame * U)
        # ---- noise: `back` frames of context behind the piece, `look` ahead
        mags_now, _ = self._rows(self.nkeys[0], sl, vm)
        hist = 0 if self._prev is None else self._prev['mags'].shape[1]
        z_now = self._take_noise(B, P, (nb + look) * U, vm, dev)              # [R, (nb + look) U]
        if hist:
            mags_ctx = torch.cat([self._prev['mags'], mags_now], dim=1)
            z_ctx = torch.cat([self._prev['noise'], z_now], dim=1)
        else:
            mags_ctx, z_ctx = mags_now, z_now
        keep = min(back, hist + nb)
        self._prev = None if keep == 0 else {
            'mags': mags_ctx[:, hist + nb - keep:hist + nb].contiguous(),
            'noise': z_ctx[:, (hist + nb - keep) * U:(hist + nb) * U].contiguous()}
        nz = self.noise
        sig = nz.get_signal(nz.get_controls(mags_ctx.contiguous())['magnitudes'], noise=z_ctx.contiguous())
        sig = sig[:, hist * U:hist * U + n].contiguous()                       # [R, n]
        # ---- add chain
        dry = torch.empty((B, n), dtype=torch.float32, device=dev)
        _lib.check(_lib_().ddspp_mix_voices(_ptr(mix.contiguous()), 1, _ptr(sig), P, _ptr(dry), B, n, n, 1 if vm else 0,
                                            _stream()))
        # ---- drop the rendered frames (and their noise samples; the look-ahead frame stays)
        self._buf = {k: v[:, nb:] for k, v in self._buf.items()}
        if self._noise_buf is not None:
            self._noise_buf = self._noise_buf[:, :, n:]
        self.frame += nb
        return self._reverb(dry)

    def _take_noise(self, B, P, n, vm, dev):
        """Noise rows [R, n] for absolute samples [frame * U, frame * U + n): explicit rows when the caller pushed them,
        else the library's Philox stream addressed by (row, absolute sample)."""
        R = B * P
        if self._noise_buf is not None:
            z = self._noise_buf[:, :, :n]
            if z.shape[2] < n:
                raise ValueError('pushed fewer noise samples than control frames')
            return (z.transpose(0, 1) if vm else z).contiguous().reshape(R, n)
        seed = getattr(self.noise, 'seed', 0)
        pos = self.frame * self.U
        if n % 4 or pos % 4:
            raise ValueError('the library noise stream is addressed in blocks of 4 samples: the hop must be a multiple of 4')
        out = torch.empty((R, n), dtype=torch.float32, device=dev)
        _lib.check(_lib_().ddspp_uniform_noise_rows(_ptr(out), R, n, int(seed) & (2 ** 64 - 1), pos // 4, 1 << 34, _stream()))
        return out

    def _reverb(self, dry):
        rv = self.reverb
        if rv is None:
            return dry
        n = dry.shape[1]
        if isinstance(rv, FeedbackDelayNetwork) and rv.trainable:
            ir = rv.get_controls(dry)['ir']
        else:
            if self._ir is None:
                raise ValueError('the reverb needs its impulse response: pass %r with the first push' % self.rkeys)
            ir = self._ir
        L = int(ir.shape[-1])
        x = dry if self._tail is None else torch.cat([self._tail, dry], dim=1)
        y = rv.get_signal(x.contiguous(), ir)[:, x.shape[1] - n:]
        self._tail = x[:, max(0, x.shape[1] - (L - 1)):].contiguous()
        return y.contiguous()


def render_range(make_synth, features, frame_lo, frame_hi, noise=None):
    """Audio [B, (frame_hi - frame_lo) * U] of frames [frame_lo, frame_hi) of a file given as whole-file features
    ({key_i: [B, T, C]} + reverb controls), rendered from scratch: the unit of work of a time shard.  frame_lo must be
    a multiple of the block size.  ``make_synth()`` returns a fresh StreamingSynthesizer; noise [B, P, T * U] or None."""
    syn = make_synth()
    U, blk = syn.U, syn.block
    if frame_lo % blk:
        raise ValueError(f'frame_lo must be a multiple of {blk} frames (a 1000-sample chunk boundary)')
    ctl = {k: core.tf_float32(v) for k, v in features.items() if k not in syn.rkeys}
    T = next(iter(ctl.values())).shape[1]
    frame_hi = min(int(frame_hi), T)
    # reverb history: render L - 1 samples early (whole blocks), discard them
    halo = 0
    if syn.reverb is not None:
        L = int(features[syn.rkeys[0]].shape[-1]) if syn.rkeys and syn.rkeys[0] in features else int(2 * syn.additive.sample_rate)
        halo = int(math.ceil(math.ceil((L - 1) / U) / blk)) * blk
    start = max(0, frame_lo - halo)
    state, rows_vm = None, None               # the row order (voice / segment major) of everything handed to `syn` below
    if start > 0:
        # phase-only pass over the prefix: the state every oscillator has at frame `start`
        P = syn.P
        f0, vm = _stack_voices([ctl[f'{syn.akeys[3]}_{i}'][:, :start + 1].contiguous() for i in range(P)])
        inh, _ = _stack_voices([ctl[f'{syn.akeys[2]}_{i}'][:, :start + 1].contiguous() for i in range(P)], vm)
        rows_vm = vm
        H = ctl[f'{syn.akeys[1]}_0'].shape[-1]
        state = core.oscillator_phase_state(f0, start * U // 1000, U, syn.additive.sample_rate,
                                            inharm_coef=inh.reshape(inh.shape[0], -1).contiguous(), n_harmonics=H)
    syn.reset(frame=start, phase_state=state)
    syn._vm = rows_vm
    back, ahead = noise_reach(ctl[f'{syn.nkeys[0]}_0'].shape[-1], getattr(syn.noise, 'window_size', 257), U)
    c0 = max(0, start - back)
    if start > c0:                              # the frames before `start` that the noise FIR reaches
        P = syn.P
        mags, vm = _stack_voices([ctl[f'{syn.nkeys[0]}_{i}'][:, c0:start].contiguous() for i in range(P)], syn._vm)
        syn._vm = vm
        B = mags.shape[0] // P
        if noise is not None:
            z = core.tf_float32(noise)[:, :, c0 * U:start * U]
            zr = (z.transpose(0, 1) if vm else z).contiguous().reshape(B * P, (start - c0) * U)
        else:
            syn.frame = c0
            zr = syn._take_noise(B, P, (start - c0) * U, vm, mags.device)
            syn.frame = start
        syn._prev = {'mags': mags.contiguous(), 'noise': zr}
    last = frame_hi >= T
    stop = T if last else min(T, frame_hi + max(1, ahead))      # look-ahead: oscillators one frame, noise FIR `ahead`
    piece = {k: v[:, start:stop] for k, v in ctl.items()}
    for k in syn.rkeys:
        if k in features:
            piece[k] = features[k]
    z = None if noise is None else core.tf_float32(noise)[:, :, start * U:stop * U]
    # a range that is not the file's last one but whose look-ahead reaches the end of the controls has nothing more
    # to wait for either: without `final` push() would hold the look-ahead frames back and come up short
    out = syn.push(piece, noise=z, final=last or stop >= T)
    need = (frame_hi - start) * U
    if out.shape[1] < need:
        raise ValueError(f'frames [{frame_lo}, {frame_hi}) do not end on a block boundary of {blk} frames')
    return out[:, (frame_lo - start) * U:need].contiguous()
