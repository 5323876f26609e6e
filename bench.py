#!/usr/bin/env python3
"""bench.py -- DDSP-Piano synthesis hot path on MI355X: audio samples / second, full chain.

One "step" = one pass of the polyphonic ProcessorGroup in the reference's call form
``processor_group(features, return_outputs_dict=True)`` (ddsp_piano/modules/piano_model.py:160):
get_controls -> inharmonic oscillator bank -> FilteredNoise -> add chain -> reverb, plus the outputs
dictionary the reference builds (dry mix, last voice's stems and controls), over one batch of synthetic
control envelopes that already sit in HBM, plus -- when more than one GPU takes part -- the final RCCL
gather of the audio to rank 0 (DDSPP_BENCH_GATHER=all: to every rank).

Workload at N GPUs (weak scaling): BASELINE.json config 3 per GPU = 64 segments x 3 s, poly=16,
24 kHz, 250 Hz controls, maestro-v2 dims (H=128 harmonics, K=96 noise bands, S=1), 3 s reverb IR;
8 GPUs x 64 = the batch=512 of config 4.  The JSON line also carries
  * roofline       : the operator-boundary cos_oscillator_bank kernel (8 B read per
                     oscillator-sample + 4 B written per sample) timed with HIP events on materialised
                     [rows, N, H] envelopes of the same workload, against 8 TB/s;
  * roofline_step  : the timed step's own dominant call (the compacted oscillator bank, VALU bound): live
                     HIP-event time against the VALU issue ceiling, instruction counts from profiles/; since round 5 also
                     against the MEASURED ceiling of this run (measured_ceiling: ddspp_fma_probe, a pure multiply-add
                     stream for 20 ms -- what the chip sustains at its power limit, not the nominal 2 cycles at 2.4 GHz);
  * roofline_noise : the step's second large call, FilteredNoise: live time against the issue time of its useful
                     multiply-adds (samples x taps) and of all its VALU instructions (counts from profiles/);
  * cpu_baseline   : the float32-faithful numpy restatement (oracle/, the only runnable stand-in for the
                     TF/ddsp reference here) and an op-by-op torch-CPU version on all cores, timed on this host;
  * value          : global batch x samples / MEDIAN of the per-step device times (every step bracketed by HIP events
                     on the launch stream and synchronised; with N > 1 the step includes its gather, max over ranks)
                     -- SURVEY.md 8(d)'s definition.  `pipelined` holds the wall clock of K back-to-back steps (barrier +
                     synchronize on both sides, max over ranks), where the side stream / the gather of one step overlap
                     the next;
  * roofline.measured_peak : device copy / triad / read rates measured in the same run, next to the 8 TB/s spec;
  * counters_stale : true when profiles/step_valu.json / osc_traffic.json describe another build of the kernels
                     (their csrc hash differs): frac_mix / traffic are then dropped;
  * step_ms        : per-step HIP-event times (median / min / max) of the headline call;
  * sustained      : >= 5 s of back-to-back headline steps (one HIP event per step, no synchronisation inside), the median
                     of the LAST HALF of the per-step times, with the shader clock and the socket power sampled meanwhile
                     (round 5: the K timed steps are 30 ms of GPU time -- this is what the step costs once power and
                     thermals have settled);
  * audio_only_call, all_stems_call, decompose_call, dense_worst_case, moving_f0, single_stream, c5_per_gpu_share, c5_full
    (BASELINE config 5 at its STATED batch of 256 on this one GPU, peak HBM use next to it), dafx22_dims, default_model_dag,
    shipped_configs (every gin file of the reference; the ENSTDkCl rows with the FDN impulse response kept AND designed per
    call): other call forms / inputs.  --extras full adds the hipGraph / one-call-driver forms of a single stream, the
    whole-file renders (136 s, 20 min) and the torch-CPU leg of cpu_baseline; --extras none drops them all;
  * phase_seconds  : wall clock of the sections of this script (the default run is sized to finish in well under a minute
                     after `import torch`, so that the driver's GPU-busy sampler sees the GPU at work);
  * with N > 1: per_rank (every rank's own median step and pipelined step), gather_to_rank0 / allgather alone, and
    gather_hidden (pipelined step against the synchronised one minus the collective: does the async gather hide?).
Launch: python bench.py [--gpus N --steps K --warmup W].  With N > 1 and no torchrun environment the script
starts the N ranks itself (torch.distributed.run, one process per GPU) and fails loudly when the box has fewer
than N GPUs.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_BYTES = 8.0e12          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
N_SIMDS = 1024                   # 256 CUs x 4 SIMDs
VALU_CYCLES_PER_WAVE_INST = 2    # a wave64 VALU instruction issues over 2 cycles on a SIMD-32 (MI355X_MICROARCH.md)
MAX_CLOCK_HZ = 2.4e9


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=64, help='segments per GPU')
    ap.add_argument('--poly', type=int, default=16)
    ap.add_argument('--harmonics', type=int, default=128)
    ap.add_argument('--bands', type=int, default=96)
    ap.add_argument('--substrings', type=int, default=1)
    ap.add_argument('--seconds', type=float, default=3.0)
    ap.add_argument('--sample-rate', type=int, default=24000)
    ap.add_argument('--ir-seconds', type=float, default=3.0)
    ap.add_argument('--roofline-rows', type=int, default=0, help='rows (segments x voices) of the '
                    'materialised oscillator-bank measurement; 0 = all rows that fit')
    ap.add_argument('--call-form', choices=['outputs_dict', 'audio_only'], default='outputs_dict',
                    help='the call the timed step makes: the reference\'s (piano_model.py:160) or group(features)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--extras', choices=['default', 'full', 'none'], default='default',
                    help='the other call forms / workloads reported next to the headline: the quick set, everything, nothing')
    ap.add_argument('--no-extras', '--no-single-stream', dest='no_extras', action='store_true',
                    help='same as --extras none')
    ap.add_argument('--sustain-seconds', type=float, default=5.0,
                    help='length of the `sustained` measurement (0: skip)')
    ap.add_argument('--cpu-voices', type=int, default=16)
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------
# launching N ranks (the driver uses torch.distributed.run itself; a bare `python bench.py --gpus N` must
# not silently measure one GPU)
# ----------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launcher_command(gpus, env, argv, device_count):
    """What `python bench.py --gpus N` has to do given the environment.

    Returns None when this process is a rank (or N == 1), the torch.distributed.run command line that
    starts N ranks of this script otherwise; raises SystemExit when N ranks cannot exist here."""
    if gpus < 1:
        raise SystemExit(f'bench.py: --gpus {gpus} is not a GPU count')
    if 'WORLD_SIZE' in env:
        world = int(env['WORLD_SIZE'])
        if world != gpus:
            raise SystemExit(f'bench.py: --gpus {gpus} but the launcher started WORLD_SIZE={world} ranks; '
                             'refusing to report a number for a different GPU count')
        return None
    if gpus == 1:
        return None
    if device_count < gpus and env.get('DDSPP_BENCH_SHARE_GPU') != '1':
        raise SystemExit(f'bench.py: --gpus {gpus} requested but this box has {device_count} GPU(s); '
                         'not falling back to fewer ranks')
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={gpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + list(argv)


# ----------------------------------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------------------------------
def make_features(B, P, T, H, K, S, L, device, seed, silent_frac=0.25, midi_lo=21, midi_hi=108, vibrato=0.0):
    """Synthetic post-network controls, generated on the GPU; per-voice keys are
    views of one [B, P, T, C] buffer, which is how a batched control network hands them over.

    vibrato > 0: every voice's f0 (and so every partial's frequency) moves in every frame (a 5 Hz sine of that
    relative depth plus a slow glide) -- the held-note fast paths of the oscillator bank never apply."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def randn(*shape):
        return torch.randn(*shape, generator=g, device=device, dtype=torch.float32)

    midi = torch.randint(midi_lo, midi_hi + 1, (B, P, 1, 1), generator=g, device=device).to(torch.float32)
    f0 = 440.0 * torch.pow(2.0, (midi - 69.0) / 12.0)
    silent = torch.rand(B, P, 1, 1, generator=g, device=device) < silent_frac
    f0 = torch.where(silent, torch.zeros_like(f0), f0)
    detune = torch.pow(2.0, 0.3 * torch.arange(S, device=device, dtype=torch.float32) / 1200.0)
    f0 = (f0 * detune.view(1, 1, 1, S)).expand(B, P, T, S).contiguous()
    if vibrato > 0.0:
        tt = torch.arange(T, device=device, dtype=torch.float32).view(1, 1, T, 1) / 250.0
        phi = 6.2831855 * torch.rand(B, P, 1, 1, generator=g, device=device)
        f0 = (f0 * (1.0 + vibrato * torch.sin(6.2831855 * 5.0 * tt + phi) + 0.1 * vibrato * tt)).contiguous()
    inharm = (torch.exp(-0.105 * midi - 6.87) + torch.exp(0.094 * midi - 13.70)).expand(B, P, T, 1).contiguous()
    decay = torch.exp(-torch.arange(T, device=device, dtype=torch.float32) / (0.4 * T)).view(1, 1, T, 1)
    amps = (randn(B, P, 1, 1) - 1.0) + 3.0 * (decay - 1.0)
    amps = amps.expand(B, P, T, 1).contiguous()
    hd = randn(B, P, T, H) - 0.05 * torch.arange(1, H + 1, device=device, dtype=torch.float32).view(1, 1, 1, H)
    hd = torch.nn.functional.avg_pool2d(hd.view(B * P, 1, T, H), (5, 1), stride=1, padding=(2, 0),
                                        count_include_pad=False).view(B, P, T, H).contiguous()
    mags = randn(B, P, T, K)
    n = torch.arange(L, device=device, dtype=torch.float32)
    ir = randn(B, L) * torch.exp(-6.9 * n / L).view(1, L) * 0.02
    ir[:, 1] = 3.0
    base = dict(amplitudes=amps, harmonic_distribution=hd, inharm_coef=inharm, f0_hz=f0, magnitudes=mags)
    feats = {f'{k}_{i}': v[:, i] for k, v in base.items() for i in range(P)}
    feats['reverb_ir'] = ir.contiguous()
    return feats, base


def synthetic_piano_roll(rng, T, frame_rate=250, notes_per_s=9.0):
    """A piano roll [T, 88, 2] (active, onset velocity in (0, 1]) like note_seq's sequence_to_pianoroll hands
    io_utils.load_midi_as_conditioning (io_utils.py:106-113): notes held over from before the segment, a Poisson stream of
    onsets (single notes and chords of 2-4), durations 80 ms .. 2.5 s (sustained passages overlap: polyphony 1 .. 12),
    pitches around the middle of the keyboard with occasional bass / treble notes."""
    roll = np.zeros((T, 88, 2), np.float32)

    def note(t0, dur, pitch, vel, onset=True):
        t1 = min(T, t0 + max(int(dur * frame_rate), 2))
        k = int(np.clip(pitch, 21, 108)) - 21
        if roll[t0:t1, k, 0].any():
            return
        roll[t0:t1, k, 0] = 1.0
        if onset:
            roll[t0, k, 1] = vel
    for _ in range(rng.poisson(2.5)):                        # already sounding at the start (their onsets lie before it)
        note(0, rng.uniform(0.2, 1.5), rng.normal(58, 14), 0.0, onset=False)
    t = 0.0
    while True:
        t += rng.exponential(1.0 / notes_per_s)
        t0 = int(t * frame_rate)
        if t0 >= T - 2:
            break
        n = 1 if rng.random() > 0.3 else int(rng.integers(2, 5))
        root = rng.normal(60, 13) if rng.random() > 0.15 else rng.choice([rng.uniform(21, 40), rng.uniform(88, 108)])
        for j in range(n):
            note(t0, float(np.clip(rng.lognormal(-0.9, 0.8), 0.08, 2.5)), root + (0 if j == 0 else rng.choice([3, 4, 7, 12, -12, 16])),
                 float(np.clip(rng.normal(0.55, 0.18), 0.08, 1.0)))
    return roll


def make_midi_like_features(dp, B, P, T, H, K, S, L, device, seed):
    """Controls shaped like a performance instead of 12 held notes per segment: a synthetic piano roll per
    segment through MIDIRoll2Conditioning (midi_encoders.py:33-104, the package's C++ allocator), then what the control
    networks do with its two columns -- f0 = midi_to_hz(pitch) (sub_modules.py:942: pitch 0, a free voice, is 8.18 Hz and
    gated by min_frequency), inharmonicity from the v2 tuning curve of the pitch, amplitudes re-triggered at every onset (level from
    the velocity, the bench's exponential decay counted from the onset), the rest as make_features draws it.
    Returns (features, base, stats)."""
    rng = np.random.default_rng(seed)
    pitch = np.zeros((B, P, T), np.float32)
    since = np.zeros((B, P, T), np.float32)
    level = np.zeros((B, P, T), np.float32)
    poly = []
    for b in range(B):
        enc = dp.MIDIRoll2Conditioning(P)
        cond, polyphony = enc(synthetic_piano_roll(rng, T))
        poly.append(np.asarray(polyphony))
        pit, vel = cond[:, :, 0].T, cond[:, :, 1].T                     # [P, T]
        pitch[b] = pit
        # frames since the voice's last onset (or since the segment started, for a note held over) and that onset's velocity
        started = (vel > 0) | (np.diff(pit, axis=1, prepend=0.0) != 0)
        idx = np.where(started, np.arange(T)[None, :], 0)
        last = np.maximum.accumulate(idx, axis=1)
        since[b] = np.arange(T)[None, :] - last
        v_on = np.take_along_axis(np.where(vel > 0, vel, 0.5), last, axis=1)
        level[b] = v_on
    poly = np.stack(poly)
    g = torch.Generator(device=device)
    g.manual_seed(seed)

    def randn(*shape):
        return torch.randn(*shape, generator=g, device=device, dtype=torch.float32)
    midi = torch.as_tensor(pitch, device=device).view(B, P, T, 1)
    active = midi > 0
    f0 = 440.0 * torch.pow(2.0, (midi - 69.0) / 12.0)                   # midi_to_hz; pitch 0 -> 8.1758 Hz
    detune = torch.pow(2.0, 0.3 * torch.arange(S, device=device, dtype=torch.float32) / 1200.0)
    f0 = (f0 * detune.view(1, 1, 1, S)).contiguous()
    inharm = (torch.exp(-0.105 * midi - 6.87) + torch.exp(0.094 * midi - 13.70)).contiguous()
    age = torch.as_tensor(since, device=device).view(B, P, T, 1)
    lvl = torch.as_tensor(level, device=device).view(B, P, T, 1)
    amps = (-1.0 + 3.0 * (lvl - 0.55)) + 3.0 * (torch.exp(-age / (0.4 * 750.0)) - 1.0)
    amps = torch.where(active, amps, torch.full_like(amps, -10.0)).contiguous()
    hd = randn(B, P, T, H) - 0.05 * torch.arange(1, H + 1, device=device, dtype=torch.float32).view(1, 1, 1, H)
    hd = torch.nn.functional.avg_pool2d(hd.view(B * P, 1, T, H), (5, 1), stride=1, padding=(2, 0),
                                        count_include_pad=False).view(B, P, T, H).contiguous()
    mags = randn(B, P, T, K)
    n = torch.arange(L, device=device, dtype=torch.float32)
    ir = randn(B, L) * torch.exp(-6.9 * n / L).view(1, L) * 0.02
    ir[:, 1] = 3.0
    base = dict(amplitudes=amps, harmonic_distribution=hd, inharm_coef=inharm, f0_hz=f0, magnitudes=mags)
    feats = {f'{k}_{i}': v[:, i] for k, v in base.items() for i in range(P)}
    feats['reverb_ir'] = ir.contiguous()
    onsets = int((np.diff((pitch > 0).astype(np.int8), axis=2, prepend=0) > 0).sum())
    stats = {'audible_voice_frames': float((pitch > 0).mean()), 'polyphony_mean': float(poly.mean()), 'polyphony_max': float(poly.max()),
             'voice_onsets_per_segment': onsets / B, 'voices_used_per_segment': float(((pitch > 0).any(axis=2)).sum(axis=1).mean())}
    return feats, base, stats


def build_group(dp, P, sr, frame_rate=250):
    additive = dp.MultiInharmonic(name='additive', frame_rate=frame_rate, sample_rate=sr, inference=True)
    noise = dp.DynamicSizeFilteredNoise(name='noise', frame_rate=frame_rate, sample_rate=sr)
    reverb = dp.Reverb(name='reverb', trainable=False)
    dag = dp.polyphonic_dag(additive, noise, reverb,
                            additive_controls=['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz'],
                            noise_controls=['magnitudes'], reverb_controls=['reverb_ir'], n_synths=P)
    return dp.ProcessorGroup(dag)


def build_default_model_group(dp, P, sr, frame_rate=250):
    """The node list of ddsp_piano/default_model.py:44-80 (noise node first, explicit ddsp.processors.Add nodes)."""
    noise = dp.DynamicSizeFilteredNoise(name='noise', frame_rate=frame_rate, sample_rate=sr)
    additive = dp.MultiInharmonic(name='additive', frame_rate=frame_rate, sample_rate=sr, inference=True)
    ctl = ['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz']
    dag = [(noise, ['magnitudes_0']), (additive, [c + '_0' for c in ctl]),
           (dp.Add(name='add_0'), ['noise/signal', 'additive/signal'])]
    for i in range(1, P):
        dag.append((additive, [c + f'_{i}' for c in ctl]))
        dag.append((noise, [f'magnitudes_{i}']))
        dag.append((dp.Add(name=f'sub_add_{i}'), ['noise/signal', 'additive/signal']))
        dag.append((dp.Add(name=f'add_{i}'), [f'add_{i - 1}/signal', f'sub_add_{i}/signal']))
    dag.append((dp.Reverb(name='reverb', trainable=False), [f'add_{P - 1}/signal', 'reverb_ir']))
    return dp.ProcessorGroup(dag)


def build_shipped_group(dp, cfg, P, sr, frame_rate=250):
    """The processor group of one of the reference's shipped gin files: the flags that reach the
    hot path, the reverb node the file names."""
    ctl = ['amplitudes', 'harmonic_distribution', 'inharm_coef', 'f0_hz']
    reverb_controls = ['reverb_ir']
    if cfg in ('ENSTDkCl-8kHz', 'ENSTDkCl-32kHz'):       # exp_tanh, no re-normalisation, an FDN that holds its parameters
        additive = dp.MultiInharmonic(name='additive', frame_rate=frame_rate, sample_rate=sr, inference=True,
                                      scale_fn=dp.exp_tanh, normalize_after_nyquist_cut=False)
        noise = dp.DynamicSizeFilteredNoise(name='noise', frame_rate=frame_rate, sample_rate=sr, scale_fn=dp.exp_tanh)
        reverb = dp.FeedbackDelayNetwork(trainable=True, delay_trainable=True, delay_lines=8 if sr == 8000 else 6,
                                         sampling_rate=sr, name='fdn', seed=11)
        reverb_controls = []
    elif cfg == 'multi_instruments':
        additive = dp.MultiInharmonic(name='additive', frame_rate=frame_rate, sample_rate=sr, inference=True,
                                      scale_fn=dp.exp_tanh, normalize_after_nyquist_cut=False)
        noise = dp.DynamicSizeFilteredNoise(name='noise', frame_rate=frame_rate, sample_rate=sr, scale_fn=dp.exp_tanh)
        reverb = dp.Reverb(name='reverb', trainable=False, add_dry=False)
    elif cfg == 'surrogate':
        additive = dp.SurrogateAdditive(name='additive', frame_rate=frame_rate, sample_rate=sr, inference=True,
                                        scale_fn=dp.exp_tanh, normalize_harm_distribution=False)
        noise = dp.DynamicSizeFilteredNoise(name='noise', frame_rate=frame_rate, sample_rate=sr, scale_fn=dp.exp_tanh)
        reverb = dp.Reverb(name='reverb', trainable=False)
        ctl = ['amplitudes', 'decays', 'decay_time', 'harmonic_distribution', 'inharm_coef', 'f0_hz']
    else:                                                # maestro-v2, dafx22-24kHz: the defaults
        additive = dp.MultiInharmonic(name='additive', frame_rate=frame_rate, sample_rate=sr, inference=True)
        noise = dp.DynamicSizeFilteredNoise(name='noise', frame_rate=frame_rate, sample_rate=sr)
        reverb = dp.Reverb(name='reverb', trainable=False)
    dag = dp.polyphonic_dag(additive, noise, reverb, additive_controls=ctl, noise_controls=['magnitudes'],
                            reverb_controls=reverb_controls, n_synths=P)
    return dp.ProcessorGroup(dag)


# ----------------------------------------------------------------------------------------------------
# timing
# ----------------------------------------------------------------------------------------------------
def time_steps(fn, steps, warmup, dist=None, drain=None):
    for _ in range(warmup):
        fn()
    if drain is not None:
        drain()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if drain is not None:
        drain()                       # the last step's collective is inside the timed region
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    return time.perf_counter() - t0


def event_times(fn, reps, warmup=2):
    """Per-call device times (ms) of fn on torch's current stream (the library launches there), HIP events."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        fn()
        e1.record(stream)
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return ts


def ms_summary(ts):
    return {'median': float(np.median(ts)), 'min': float(np.min(ts)), 'max': float(np.max(ts)), 'n': len(ts)}


class GpuSampler:
    """Shader clock (MHz) and socket power (W) of one GPU, read from a thread while a measurement runs: the amdgpu hwmon
    files when they are readable (cheap), `rocm-smi --showpower --showclocks --json` otherwise."""

    def __init__(self, index, period=0.25):
        import glob
        import threading
        self.index, self.period = index, period
        self.samples, self._stop, self._thread = [], threading.Event(), None
        self.source, self._power, self._freq = None, None, None
        cards = []
        for hw in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')):
            pw = next((os.path.join(hw, f) for f in ('power1_average', 'power1_input') if os.path.exists(os.path.join(hw, f))), None)
            fq = os.path.join(hw, 'freq1_input')
            if pw and os.path.exists(fq):
                try:
                    slot = open(os.path.join(hw, '..', '..', 'uevent')).read()
                    slot = next((ln.split('=', 1)[1] for ln in slot.splitlines() if ln.startswith('PCI_SLOT_NAME=')), hw)
                except OSError:
                    slot = hw
                cards.append((slot, pw, fq))
        cards.sort()
        # the card of THIS torch device: by PCI address (a box may show more cards in sysfs than HIP exposes)
        want = None
        try:
            pr = torch.cuda.get_device_properties(index)
            want = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
        except Exception:  # noqa: BLE001
            pass
        pick = next((c for c in cards if want and c[0].lower() == want), None)
        if pick is None and want is None and index < len(cards):
            pick = cards[index]
        if pick is not None:
            try:
                float(open(pick[1]).read())
                float(open(pick[2]).read())
                self._power, self._freq, self.source = pick[1], pick[2], 'sysfs hwmon ' + pick[0]
            except (OSError, ValueError):
                pass
        self.cards_seen = [c[0] for c in cards]
        if self.source is None:
            self.source = 'rocm-smi'

    def _read(self):
        if self._power:
            return float(open(self._freq).read()) * 1e-6, float(open(self._power).read()) * 1e-6
        import re
        out = subprocess.run(['rocm-smi', '-d', str(self.index), '--showpower', '--showclocks', '--json'],
                             capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(out).values()))
        clk = next((float(re.search(r'(\d+)\s*Mhz', str(v), re.I).group(1)) for k, v in card.items()
                    if 'sclk' in k.lower() and re.search(r'(\d+)\s*Mhz', str(v), re.I)), float('nan'))
        pwr = next((float(v) for k, v in card.items() if 'power' in k.lower() and '(w)' in k.lower()), float('nan'))
        return clk, pwr

    def _run(self):
        while not self._stop.is_set():
            t = time.perf_counter()
            try:
                clk, pwr = self._read()
                self.samples.append((t, clk, pwr))
            except Exception:  # noqa: BLE001  (a sampler that cannot read reports nothing; the timing does not depend on it)
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        import threading
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=15)
        return False

    def summary(self, t_from=None):
        rows = [r for r in self.samples if t_from is None or r[0] >= t_from]
        out = {'source': self.source, 'n': len(rows)}
        if t_from is None and len(getattr(self, 'cards_seen', [])) > 1:
            out['cards_in_sysfs'] = len(self.cards_seen)
        for name, col, unit in (('sclk_mhz', 1, 'MHz'), ('socket_power_w', 2, 'W')):
            vals = [r[col] for r in rows if r[col] == r[col]]
            if vals:
                out[name] = {'median': float(np.median(vals)), 'min': float(np.min(vals)), 'max': float(np.max(vals))}
        return out


def measure_sustained(step_fn, seconds, est_ms, device_index, units_per_step, sr, drain=None):
    """`seconds` of back-to-back steps: one HIP event after every step on the launch stream, nothing synchronised inside
    the loop (the host runs ahead of the GPU as far as the launch queue lets it), clock and power sampled from a thread.
    Reported: the median of the LAST HALF of the per-step times -- the state the GPU settles in -- next to the first half's."""
    n = int(min(20000, max(200, seconds * 1e3 / max(est_ms, 1e-3) * 1.05)))
    stream = torch.cuda.current_stream()
    events = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    with GpuSampler(device_index) as smp:
        t0 = time.perf_counter()
        events[0].record(stream)
        for i in range(n):
            step_fn()
            events[i + 1].record(stream)
        if drain is not None:
            drain()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    ts = np.asarray([events[i].elapsed_time(events[i + 1]) for i in range(n)])
    half = ts[n // 2:]
    med = float(np.median(half))
    return {'workload': f'{n} headline steps back to back ({t1 - t0:.2f} s of wall clock), one HIP event per step, no '
                        'synchronisation inside the loop',
            'ms_per_step': med, 'value': units_per_step / (med * 1e-3), 'rtf': units_per_step / (med * 1e-3) / sr,
            'last_half': ms_summary(half), 'first_half': ms_summary(ts[:n // 2]),
            'first_100_median': float(np.median(ts[:100])), 'last_100_median': float(np.median(ts[-100:])),
            'wall_ms_per_step': (t1 - t0) / n * 1e3, 'seconds': t1 - t0, 'n': n,
            'gpu': smp.summary(), 'gpu_last_half': smp.summary(t_from=t0 + 0.5 * (t1 - t0))}


def measure_device_peaks(device, nbytes=1 << 30):
    """HBM rates this device reaches in THIS run (GB/s, bytes read + bytes written): a device copy, a triad, a read."""
    n = nbytes // 4
    a = torch.ones(n, dtype=torch.float32, device=device)
    b = torch.full((n,), 2.0, dtype=torch.float32, device=device)
    c = torch.empty(n, dtype=torch.float32, device=device)
    copy = float(np.min(event_times(lambda: c.copy_(a), 6, warmup=2))) * 1e-3
    triad = float(np.min(event_times(lambda: torch.add(a, b, alpha=3.0, out=c), 6, warmup=2))) * 1e-3
    read = float(np.min(event_times(lambda: a.sum(), 6, warmup=2))) * 1e-3
    out = {'copy': 2 * nbytes / copy / 1e9, 'triad': 3 * nbytes / triad / 1e9, 'read': nbytes / read / 1e9,
           'bytes': nbytes, 'note': 'torch copy_ (hipMemcpyDtoD-class kernel), a + 3 b -> c, sum(a) on 1 GiB float32 buffers; '
                                    'best of 6, HIP events'}
    del a, b, c
    torch.cuda.empty_cache()
    return out


def measure_roofline(dp, base, args, T, U, device):
    """cos_oscillator_bank at the operator boundary on materialised envelopes (HBM-bound kernel)."""
    from ddsp_piano_amd import core
    B, P = base['f0_hz'].shape[:2]
    H = base['harmonic_distribution'].shape[-1]
    N = T * U
    rows_all = B * P
    free, _ = torch.cuda.mem_get_info()
    per_row = 2 * N * H * 4 + N * 4
    rows_fit = int(free * 0.85 // per_row)
    rows = min(rows_all, rows_fit) if args.roofline_rows <= 0 else min(args.roofline_rows, rows_all, rows_fit)
    if rows < 1:
        return None
    additive = dp.MultiInharmonic(sample_rate=args.sample_rate, inference=True)
    sl = slice(0, rows)
    ctl = additive._controls(base['amplitudes'].reshape(rows_all, T, 1)[sl],
                             base['harmonic_distribution'].reshape(rows_all, T, H)[sl],
                             base['inharm_coef'].reshape(rows_all, T, 1)[sl],
                             base['f0_hz'].reshape(rows_all, T, -1)[sl][..., :1].contiguous())
    hf = core.get_harmonic_frequencies(ctl['f0_hz'], H) * (1.0 + ctl['harmonic_shifts'])
    ha = ctl['amplitudes'] * ctl['harmonic_distribution']
    hf, ha = hf.contiguous(), ha.contiguous()
    fe = core.resample(hf, N)
    ae = core.resample(ha, N, method='window')
    out = torch.empty((rows, N), dtype=torch.float32, device=device)
    ws, nbytes = core._osc_workspace(rows, N, H, device)
    lib = core._lib_()

    def launch():
        rc = lib.ddspp_cos_oscillator_bank(core._ptr(fe), core._ptr(ae), core._ptr(out), rows, N, H,
                                           float(args.sample_rate), 1, 1, 1, core._ptr(ws), nbytes,
                                           core._stream())
        assert rc == 0, core._lib.last_error()

    times = [t * 1e-3 for t in event_times(launch, 5, warmup=1)]
    t = float(np.mean(times))
    alg_bytes = rows * (N * H * 8 + N * 4)
    # the ceiling for THIS kernel in THIS run: the same two envelope buffers read by the library's pure-read probe (the
    # kernel's access pattern -- a contiguous stream per wavefront, 1 KB per instruction, sixteen in flight -- and
    # nothing else), at the kernel's own stream count and at twice / four times it
    import ctypes
    sink = torch.zeros(4, dtype=torch.float32, device=device)
    probe = {}
    for waves in (2048, 4096, 8192):
        nread = ctypes.c_size_t(0)

        def read_both():
            tot = 0
            for buf in (fe, ae):
                rc = lib.ddspp_hbm_read_probe(core._ptr(buf), buf.numel(), waves, core._ptr(sink), ctypes.byref(nread), core._stream())
                assert rc == 0, core._lib.last_error()
                tot += nread.value
            return tot
        nb = read_both()
        tp = float(np.min(event_times(read_both, 4, warmup=1))) * 1e-3
        probe[waves] = nb / tp / 1e9
    read_peak = max(probe.values())
    # the reference's literal operator chain resample -> resample -> cos_oscillator_bank (inharm_synth.py:117-127) at this size:
    # the two stand-alone upsamplers (pure write streams) rewriting fe / ae in place, against the pure write of the same buffer
    lo_, hi_, w_, _ = core.linear_tables(T, N, device)
    win_ = core.hann_window(2 * U, device)

    def up_linear():
        rc = lib.ddspp_resample_linear(core._ptr(hf), core._ptr(lo_), core._ptr(hi_), core._ptr(w_), core._ptr(fe), rows, T, H, N,
                                       core._stream())
        assert rc == 0, core._lib.last_error()

    def up_window():
        rc = lib.ddspp_resample_window(core._ptr(ha), core._ptr(win_), core._ptr(ae), rows, T, H, U, core._stream())
        assert rc == 0, core._lib.last_error()
    fe_check = fe[0, :256].clone()
    t_lin = float(np.mean(event_times(up_linear, 3, warmup=1))) * 1e-3
    t_win = float(np.mean(event_times(up_window, 3, warmup=1))) * 1e-3
    assert torch.equal(fe_check, fe[0, :256])
    wprobe = {}
    scratch = torch.empty_like(fe)
    for nt in (1, 0):
        for waves in (2048, 4096):
            nwr = ctypes.c_size_t(0)

            def wr():
                rc = lib.ddspp_hbm_write_probe(core._ptr(scratch), scratch.numel(), waves, nt, ctypes.byref(nwr), core._stream())
                assert rc == 0, core._lib.last_error()
            wr()
            tw = float(np.min(event_times(wr, 3, warmup=1))) * 1e-3
            wprobe[f"{'nt' if nt else 'plain'}_{waves}"] = nwr.value / tw / 1e9
    del scratch
    write_peak = max(wprobe.values())
    up_bytes = rows * N * H * 4
    chain = {'workload': f'resample(linear) + resample(window) + cos_oscillator_bank on {rows} rows x {N} samples x {H} harmonics '
                         '(inharm_synth.py:117-127 as three operators, envelopes materialised in HBM)',
             'ms': {'resample_linear': t_lin * 1e3, 'resample_window': t_win * 1e3, 'cos_oscillator_bank': t * 1e3,
                    'total': (t_lin + t_win + t) * 1e3},
             'upsampler_bytes_written_each': up_bytes,
             'gb_per_s_written': {'resample_linear': up_bytes / t_lin / 1e9, 'resample_window': up_bytes / t_win / 1e9},
             'write_ceiling': {'gb_per_s_by_policy_and_streams': wprobe, 'best': write_peak,
                               'note': 'ddspp_hbm_write_probe: 16-byte stores, a contiguous stream per wavefront, on a buffer of '
                                       'the same size; best of 3'},
             'frac_of_measured_write': {'resample_linear': up_bytes / t_lin / 1e9 / write_peak,
                                        'resample_window': up_bytes / t_win / 1e9 / write_peak}}
    traffic, stale = None, None
    tf = os.path.join(ROOT, 'profiles', 'osc_traffic.json')
    if os.path.exists(tf):
        try:
            prof = json.load(open(tf))
            stale = prof.get('csrc_hash') != dp._lib.source_hash()
            # the counter pass ran at rows=1024: per-launch bytes scale with the rows of this launch
            if not stale:
                traffic = prof['hbm_bytes_per_launch'] * alg_bytes / prof['algorithmic_bytes_per_launch']
        except Exception:  # noqa: BLE001
            traffic = None
    del fe, ae, out, hf, ha
    torch.cuda.empty_cache()
    peaks = measure_device_peaks(device)
    peaks['library_read_probe'] = {'gb_per_s_by_streams': {str(k): v for k, v in probe.items()},
                                   'note': 'ddspp_hbm_read_probe on the kernel\'s own two envelope buffers '
                                           '(pure read, the kernel\'s access pattern); best of 4'}
    # measured_peak: the fastest way this run found to move these bytes -- the pure read of the same buffers, or torch's
    # copy / triad / read if one of them beats it; the kernel does the read AND arithmetic, so frac_of_measured_peak <= 1
    # up to run-to-run noise.  guide_achievable: the 6.29 TB/s float4 copy of MI355X_MICROARCH.md.
    best = max(read_peak, peaks['copy'], peaks['triad'], peaks['read'])
    return {'bound': 'hbm', 'achieved': alg_bytes / t / 1e9, 'peak': HBM_PEAK_BYTES / 1e9, 'unit': 'GB/s',
            'frac': alg_bytes / t / HBM_PEAK_BYTES, 'traffic': traffic, 'counters_stale': stale,
            'measured_peak': best, 'frac_of_measured_peak': alg_bytes / t / 1e9 / best, 'measured': peaks,
            'guide_achievable': 6290.0, 'frac_of_guide_achievable': alg_bytes / t / 1e9 / 6290.0,
            'kernel': 'ddspp::osc_stream_kernel<H / 64> (ddspp_cos_oscillator_bank on materialised envelopes, angular cumsum, '
                      'summed, spans=1: every envelope byte read once)',
            'three_operator_chain': chain,
            'rows': rows, 'n_samples': N, 'n_harmonics': H, 'algorithmic_bytes_per_launch': alg_bytes,
            'ms_per_launch': t * 1e3, 'ms_min': float(np.min(times)) * 1e3}


def measure_valu_ceiling(device, waves_per_simd=4, target_ms=20.0):
    """What the chip SUSTAINS in wave64 multiply-adds, in this run: ddspp_fma_probe (a pure stream of independent v_fmac_f32,
    `waves_per_simd` wavefronts on every SIMD) timed with HIP events over ~20 ms -- the power management needs milliseconds
    to settle, and under such a stream the clock falls well below 2.4 GHz.  Returns instructions per second over the chip."""
    import ctypes
    from ddsp_piano_amd import core
    lib = core._lib_()
    sink = torch.zeros(4, dtype=torch.float32, device=device)
    per_simd = ctypes.c_double(0.0)

    def run(iters):
        rc = lib.ddspp_fma_probe(core._ptr(sink), waves_per_simd, int(iters), ctypes.byref(per_simd), core._stream())
        assert rc == 0, core._lib.last_error()

    iters = 4000
    t = float(np.median(event_times(lambda: run(iters), 2, warmup=1)))                 # sizes the real run
    iters = max(1000, int(iters * target_ms / max(t, 1e-3)))
    t = float(np.min(event_times(lambda: run(iters), 2, warmup=0))) * 1e-3
    ns = t * 1e9 / per_simd.value
    return {'wave_instructions_per_s': N_SIMDS / (ns * 1e-9), 'ns_per_instruction_per_simd': ns, 'waves_per_simd': waves_per_simd,
            'ms': t * 1e3, 'note': 'ddspp_fma_probe: independent wave64 v_fmac_f32 (three VGPR operands) on every SIMD, '
                                   'HIP events; the power-limited rate the VALU-bound kernels are read against'}


def measure_roofline_step(dp, base, args, T, U, device, ceiling=None):
    """The timed step's dominant call, the compacted oscillator bank (ddspp_polyphonic_additive: VALU bound, it
    moves ~0.7 GB): HIP-event time of the call on the bench inputs against the VALU issue ceiling.  The wave-
    instruction count of its kernels comes from the committed counter pass (profiles/step_valu.json, written by
    tools/step_pmc.sh on the same workload); issue fraction = instructions x 2 cycles / (1024 SIMDs x clock x time)."""
    from ddsp_piano_amd import core
    B, P = base['f0_hz'].shape[:2]
    H = base['harmonic_distribution'].shape[-1]
    S = base['f0_hz'].shape[-1]
    N = T * U
    R = B * P
    additive = dp.MultiInharmonic(sample_rate=args.sample_rate, inference=True)
    ctl = additive._controls(base['amplitudes'].reshape(R, T, 1), base['harmonic_distribution'].reshape(R, T, H),
                             base['inharm_coef'].reshape(R, T, 1), base['f0_hz'].reshape(R, T, S), want_counts=True,
                             want_shifts=False)

    def launch():          # as the batched group calls it: shifts formed in the kernels from inharm_coef
        core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T), ctl['harmonic_distribution'],
                                 None, B, N, args.sample_rate, voice_major=False, audible=ctl['_audible'],
                                 inharm_coef=ctl['_inharm_coef'].reshape(R, T))

    ts = event_times(launch, 10, warmup=2)
    t = float(np.median(ts)) * 1e-3
    out = {'bound': 'valu', 'call': 'ddspp_polyphonic_additive (pre-pass + counts + compacted bank + slot sum)',
           'ms_per_call': t * 1e3, 'ms_min': float(np.min(ts)), 'unit': 'wave64 VALU instructions/s',
           'peak': N_SIMDS * MAX_CLOCK_HZ / VALU_CYCLES_PER_WAVE_INST,
           'peak_note': '1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction'}
    pf = os.path.join(ROOT, 'profiles', 'step_valu.json')
    if os.path.exists(pf):
        try:
            prof = json.load(open(pf))
            out['counters_stale'] = prof.get('csrc_hash') != dp._lib.source_hash()
            if out['counters_stale']:          # the counts describe another build of the kernels: no fractions from them
                return out
            insts = float(prof['valu_wave_instructions_per_call'])
            out.update({'valu_wave_instructions': insts, 'achieved': insts / t,
                        'frac': insts / t / out['peak'], 'counters': prof.get('source')})
            trans = float(prof.get('valu_trans_wave_instructions_per_call', 0.0))
            if trans and ceiling:
                # ... and against what the chip sustains (measure_valu_ceiling, this run): the kernels' instructions with a
                # quarter-rate transcendental counted as four plain ones
                out['measured_ceiling'] = ceiling
                out['frac_of_measured_ceiling'] = ((insts - trans) + 4.0 * trans) / t / ceiling['wave_instructions_per_s']
            if trans:
                # a v_cos_f32 / v_sqrt_f32 / v_rcp_f32 occupies the SIMD for 8 cycles, not 2: the share of the call's
                # time its instruction mix accounts for at full issue rate (1.0 = nothing but issue cycles)
                out.update({'valu_trans_wave_instructions': trans,
                            'frac_mix': ((insts - trans) * 2.0 + trans * 8.0) / (N_SIMDS * MAX_CLOCK_HZ * t),
                            'frac_mix_note': 'issue cycles with transcendentals at 8 cycles / (1024 SIMDs x 2.4 GHz x time)'})
        except Exception:  # noqa: BLE001
            pass
    return out


def measure_roofline_noise(dp, base, args, T, U, device, ceiling=None):
    """The step's second large kernel, FilteredNoise (the fused design + time-varying FIR kernel: VALU bound).  Live HIP-
    event time of the call as the batched group makes it (voice sums of 8, the last voice kept apart, scale_fn inside),
    against the issue time of (a) its USEFUL multiply-adds -- samples x taps of the windowed impulse response, what any
    direct-form implementation has to issue -- and (b) every VALU instruction the kernel executes (profiles/step_valu.json,
    dropped when that file describes another build)."""
    from ddsp_piano_amd import core
    B, P = base['f0_hz'].shape[:2]
    K = base['magnitudes'].shape[-1]
    N, R = T * U, B * P
    synth = dp.DynamicSizeFilteredNoise(sample_rate=args.sample_rate, frame_rate=250)
    mags = base['magnitudes'].reshape(R, T, K)
    x = core.uniform_noise((R, N), seed=7, device=device)
    vq = next(v for v in (8, 4, 2, 1) if P % v == 0)
    rs = synth.raw_scale()

    def launch():
        if vq > 1 and core.frequency_filter_voice_sums(x, mags, synth.window_size, rs, P, vq, False, split_last=True) is not None:
            return
        core.frequency_filter(x, mags, window_size=synth.window_size, raw_scale=rs)

    ts = event_times(launch, 10, warmup=2)
    t = float(np.median(ts)) * 1e-3
    taps = 2 * (K - 1) if synth.window_size <= 0 else min(synth.window_size, 2 * (K - 1))        # windowed impulse response
    useful = R * N * float(taps) / 64.0                                   # wave64 FMAs
    peak = N_SIMDS * MAX_CLOCK_HZ / VALU_CYCLES_PER_WAVE_INST
    out = {'bound': 'valu', 'call': 'FilteredNoise: frequency_filter over all voices (fused design + time-varying FIR)',
           'ms_per_call': t * 1e3, 'ms_min': float(np.min(ts)), 'unit': 'wave64 VALU instructions/s', 'peak': peak,
           'peak_note': '1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction',
           'useful_fma_wave_instructions': useful, 'taps': int(taps), 'frac_useful': useful / t / peak}
    pf = os.path.join(ROOT, 'profiles', 'step_valu.json')
    if os.path.exists(pf):
        try:
            prof = json.load(open(pf))
            out['counters_stale'] = prof.get('csrc_hash') != dp._lib.source_hash()
            nz = prof.get('noise')
            if nz and not out['counters_stale'] and ceiling:
                out['measured_ceiling'] = ceiling
                out['frac_of_measured_ceiling'] = nz['valu'] / t / ceiling['wave_instructions_per_s']
                out['frac_useful_of_measured_ceiling'] = useful / t / ceiling['wave_instructions_per_s']
            if nz and not out['counters_stale']:
                out.update({'valu_wave_instructions': nz['valu'], 'fma_wave_instructions': nz['fma'],
                            'mfma_mops': nz.get('mfma_mops'), 'lds_instructions': nz.get('lds'),
                            'achieved': nz['valu'] / t, 'frac': nz['valu'] / t / peak, 'kernels': nz.get('kernels')})
        except Exception:  # noqa: BLE001
            pass
    return out


def measure_cpu_baseline(args, T, U, full=False):
    """(full=False, the default run: the numpy oracle on 32 threads only, ~6 s; full=True adds a second thread count and the
    torch-CPU leg, whose 256-thread probe alone is half a minute on the GPU box's host.)
    The oracle (float32-faithful numpy restatement of the TF/ddsp reference -- TF itself cannot be
    installed here) on a bounded sample of the same workload: whole 3 s, poly-16 segments, one
    thread per voice task, sized for roughly 10 s of wall clock on this host; and the op-by-op torch-CPU
    chain (oracle/torch_cpu_chain.py) with torch's intra-op pool on all cores, the stand-in for TF's."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import ddsp_oracle as O
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import synth_controls, synth_ir
    rng = np.random.default_rng(20240)
    H, K, S, sr = args.harmonics, args.bands, args.substrings, args.sample_rate
    P = args.cpu_voices
    N = T * U
    L = int(args.ir_seconds * sr)
    additive = O.MultiInharmonic(frame_rate=250, sample_rate=sr, inference=True)
    noise = O.FilteredNoise(frame_rate=250, sample_rate=sr)
    reverb = O.Reverb()
    threads = max(1, min(os.cpu_count() or 1, 32))

    def voice(task):
        c, z = task
        a = additive(c['amplitudes'], c['harmonic_distribution'], c['inharm_coef'], c['f0_hz'])
        n = noise.get_signal(**noise.get_controls(c['magnitudes']), noise=z)
        return a, n

    def make_segment():
        return ([(synth_controls(rng, 1, T, H, S=S, K=K), rng.uniform(-1, 1, [1, N]).astype(np.float32))
                 for _ in range(P)], synth_ir(rng, 1, L))

    def run(segments, threads=threads):
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=threads) as ex:
            sigs = list(ex.map(voice, [t for seg, _ in segments for t in seg]))
        outs = []
        for si, (_, ir) in enumerate(segments):
            mix = None
            for a, z in sigs[si * P:(si + 1) * P]:
                mix = (z + a) if mix is None else ((mix + z) + a)
            outs.append(reverb.get_signal(mix, ir))
        assert outs[0].shape == (1, N)
        return time.perf_counter() - t0

    t1 = run([make_segment()])                                   # also warms numpy / scipy up
    n_seg = int(max(1, min(32, round((8.0 if full else 6.0) / max(t1, 1e-3)))))
    segs = [make_segment() for _ in range(n_seg)]
    # one thread per voice task scales as far as the tasks and the host's cores go (numpy releases the GIL in its inner
    # loops): the sample is timed with 32 threads and with as many as it has tasks (up to 128), the faster one is reported
    tried = {}
    for nthr in sorted({threads, max(1, min(os.cpu_count() or 1, 128, n_seg * P))} if full else {threads}):
        tried[nthr] = run(segs, nthr)
    threads = min(tried, key=tried.get)
    dt = tried[threads]
    numpy_port = {'value': n_seg * N / dt, 'unit': 'audio samples/s', 'cores': threads, 'kind': 'port',
                  'sample': f'{n_seg} segment(s) x {args.seconds:g} s, poly={P}, H={H}, K={K}, S={S}, {sr} Hz, full '
                            f'chain; numpy oracle, {threads} threads over voice tasks, {dt:.1f} s of wall clock '
                            f'(host has {os.cpu_count()} logical cores; tried ' +
                            ', '.join(f'{k} threads {v:.1f} s' for k, v in tried.items()) + ')',
                  'rtf': n_seg * N / dt / sr}

    if not full:
        out = dict(numpy_port)
        out['numpy_oracle'] = numpy_port
        out['note'] = ('TensorFlow / ddsp are not installable on this host: the float32-faithful numpy restatement of the '
                       'reference chain stands in (bench.py --extras full adds the op-by-op torch-CPU chain)')
        return out
    # torch-CPU: whole segments through the vectorised operator sequence, intra-op pool on all cores (and on 32, where
    # oversubscribing tiny ops hurts less; the faster of the two is reported).  Bounded: a probe on one voice x 0.5 s
    # sizes the sample to ~8 s of wall clock.
    from oracle import torch_cpu_chain as TC
    cores = os.cpu_count() or 1

    def torch_run(bt, voices_n, frames):
        n = frames * U
        voices = [{k: torch.as_tensor(v) for k, v in synth_controls(rng, bt, frames, H, S=S, K=K).items()}
                  for _ in range(voices_n)]
        noises = [torch.as_tensor(rng.uniform(-1, 1, [bt, n]).astype(np.float32)) for _ in range(voices_n)]
        ir = torch.as_tensor(synth_ir(rng, bt, L))
        t0 = time.perf_counter()
        with torch.no_grad():
            out = TC.synthesize(voices, ir, noises, sr)
        assert tuple(out.shape) == (bt, n)
        return time.perf_counter() - t0

    # sizing probe: one voice x 0.5 s, the median of three runs per thread count; a probe the clock cannot resolve
    # (< 10 ms) is repeated on a four times longer sample
    probes, probe_frames = {}, 125
    for nthr in sorted({cores, min(cores, 32)}, reverse=True):
        torch.set_num_threads(nthr)
        torch_run(1, 1, probe_frames)                          # warm-up (thread pool, FFT plans)
        probes[nthr] = float(np.median([torch_run(1, 1, probe_frames) for _ in range(3)]))
    if min(probes.values()) < 0.010:
        probe_frames *= 4
        for nthr in list(probes):
            torch.set_num_threads(nthr)
            torch_run(1, 1, probe_frames)
            probes[nthr] = float(np.median([torch_run(1, 1, probe_frames) for _ in range(3)]))
    nthr = min(probes, key=probes.get)
    torch.set_num_threads(nthr)
    per_voice_second = probes[nthr] / (probe_frames / 250.0)
    budget = 8.0
    pv = int(max(1, min(P, budget / max(per_voice_second * args.seconds, 1e-3))))      # voices of one full-length segment
    dtt = torch_run(1, pv, T)
    # throughput in segment-equivalents: pv of the P voices of a segment were synthesised
    torch_cpu = {'value': (pv / P) * N / dtt, 'unit': 'audio samples/s', 'cores': nthr, 'kind': 'port',
                 'sample': f'{pv} of the {P} voices of one {args.seconds:g} s segment (H={H}, K={K}, S={S}, {sr} Hz) + reverb; '
                           f'op-by-op torch-CPU chain (materialised envelopes, framed FFT noise, FFT reverb), intra-op pool = '
                           f'{nthr} threads (probe: {", ".join(f"{k} threads {v * 1e3:.1f} ms" for k, v in probes.items())} per voice x {probe_frames / 250:g} s, median of 3), '
                           f'{dtt:.1f} s of wall clock; value scaled to whole poly-{P} segments',
                 'rtf': (pv / P) * N / dtt / sr}
    best = max((numpy_port, torch_cpu), key=lambda d: d['value'])
    out = dict(best)
    out['numpy_oracle'] = numpy_port
    out['torch_cpu_all_cores'] = torch_cpu
    out['note'] = 'TensorFlow / ddsp are not installable on this host: both legs are restatements of the reference chain'
    return out


def main():
    args = parse()
    cmd = launcher_command(args.gpus, os.environ, sys.argv[1:], torch.cuda.device_count())
    if cmd is not None:
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: torch.cuda.is_available() is False')
    # DDSPP_BENCH_SHARE_GPU=1 (dry run of the multi-rank flow on a box with fewer GPUs than ranks: tests/test_gpu_dist.py):
    # ranks wrap around the devices and the backend must be gloo (RCCL refuses two ranks on one GPU); the line says so.
    share_gpu = os.environ.get('DDSPP_BENCH_SHARE_GPU') == '1'
    backend = os.environ.get('DDSPP_BENCH_BACKEND', 'nccl')
    dev_index = local_rank % torch.cuda.device_count() if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    dist = None
    use_dist = world > 1 or os.environ.get('DDSPP_BENCH_DIST') == '1'   # the latter: exercise RCCL with one rank
    if use_dist:
        import torch.distributed as dist_mod
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend == 'nccl':
            dist_mod.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist_mod.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_mod
        world = dist.get_world_size()            # the ranks the backend actually sees

    import ddsp_piano_amd as dp
    from ddsp_piano_amd import parallel

    extras_level = 'none' if args.no_extras else args.extras
    phase_s, _t_phase = {}, [time.perf_counter()]

    def phase_done(name):
        torch.cuda.synchronize()
        now = time.perf_counter()
        phase_s[name] = phase_s.get(name, 0.0) + now - _t_phase[0]
        _t_phase[0] = now

    sr = args.sample_rate
    U = sr // 250
    T = int(round(args.seconds * 250))
    N = T * U
    B, P, H, K, S = args.batch, args.poly, args.harmonics, args.bands, args.substrings
    L = int(args.ir_seconds * sr)
    feats, base = make_features(B, P, T, H, K, S, L, device, seed=20240 + rank)
    # The graded kernel (75 GB of envelopes) is measured FIRST, on a heap nothing has churned yet: the same launch after
    # the sections below reads 1.5 % slower (tools/roofline_order.py, four alternating pairs: 0.726-0.737 of 8 TB/s fresh
    # against 0.717-0.726 late; DESIGN.md section 4a: the rate is a property of where the process's buffers lie).
    roof = None
    if rank == 0 and not args.no_roofline:
        roof = measure_roofline(dp, base, args, T, U, device)
    phase_done('inputs_and_graded_kernel')
    pg = build_group(dp, P, sr)
    want_dict = args.call_form == 'outputs_dict'
    # the final gather: to rank 0 (the reference's strategy.gather(outputs, axis=0), evaluate_model.py:45 -- one program
    # holds the batch; DDSPP_BENCH_GATHER=all: an all-gather, every rank gets it).  In `pipelined` the gather of step i runs
    # on RCCL's stream while step i + 1 synthesises (two landing buffers)
    gather_dst = None if os.environ.get('DDSPP_BENCH_GATHER', 'rank0') == 'all' else 0
    gathered = [torch.empty((world * B, N), dtype=torch.float32, device=device) for _ in range(2)] if use_dist else None
    state = {'work': None, 'k': 0}

    def drain():
        if state['work'] is not None:
            state['work'].wait()
            state['work'] = None

    def call(group, f):
        if want_dict:
            return group(f, return_outputs_dict=True)['signal']        # piano_model.py:160-164
        return group(f)

    def step():
        audio = call(pg, feats)
        if use_dist:
            drain()
            _, state['work'] = parallel.gather_audio(audio, gathered[state['k'] & 1], async_op=True, dst=gather_dst)
            state['k'] += 1
        return audio

    for _ in range(2):                 # set-up, not a step: rocFFT plans, kernel code objects, allocator pools
        call(pg, feats)
    torch.cuda.synchronize()
    # (1) W warm-up steps, then EXACTLY K steps, each bracketed by HIP events on the launch stream and synchronised: no step
    #     overlaps its neighbours.  With N > 1 a step includes its own gather (synchronous).  value = global samples /
    #     the MEDIAN step time, max over ranks ("device-synchronised, median of >= 20 runs").
    def sync_step():
        audio = call(pg, feats)
        if use_dist:
            parallel.gather_audio(audio, gathered[0], dst=gather_dst)
        return audio

    for _ in range(args.warmup):
        sync_step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_wall0 = time.perf_counter()
    step_ts = event_times(sync_step, args.steps, warmup=0)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall_sync = time.perf_counter() - t_wall0
    med = float(np.median(step_ts)) * 1e-3
    if use_dist:
        tt = torch.tensor([med, wall_sync], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        med, wall_sync = float(tt[0].item()), float(tt[1].item())
    value = world * B * N / med
    # (2) the same K steps back to back (barrier + synchronize on both sides, max over ranks): the side stream and the
    #     gather of one step overlap the next step's kernels -- what a caller that keeps the GPU fed gets
    dt = time_steps(step, args.steps, args.warmup, dist, drain if use_dist else None)
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    extra = {'pipelined': {'ms_per_step': dt / args.steps * 1e3, 'value': world * B * N * args.steps / dt,
                           'note': f'wall clock of {args.steps} back-to-back steps (barrier + synchronize on both sides, max '
                                   'over ranks); steps overlap at their edges'},
             'synchronised_wall_ms_per_step': wall_sync / args.steps * 1e3}
    if use_dist:
        extra['backend'] = backend
        if share_gpu:
            extra['shared_gpu'] = True           # a dry run of the multi-rank flow, not a multi-GPU measurement
    if use_dist:
        # the collectives by themselves (synchronous), max over ranks: what the overlap has to hide
        audio = call(pg, feats)
        extra['gather'] = 'all ranks (all_gather_into_tensor)' if gather_dst is None else 'to rank 0 (dist.gather)'
        for name, dst in (('gather_to_rank0', 0), ('allgather', None)):
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                dist.barrier()
                t0 = time.perf_counter()
                parallel.gather_audio(audio, gathered[0], dst=dst)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            tg = torch.tensor([float(np.median(ts))], dtype=torch.float64, device=device)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            nbytes = (world - 1) * B * N * 4
            gbs = nbytes / max(float(tg.item()), 1e-9) * 1e-9
            # what the number is read against on the node: xGMI is point to point, one ~153 GB/s link per peer
            # (MI355X_MICROARCH.md: 7 links per GPU); the receiving rank takes one block from each of its world - 1 peers
            links = min(world - 1, 7)
            extra[name] = {'ms': float(tg.item()) * 1e3, 'bytes_received': nbytes, 'gb_per_s': gbs,
                           'xgmi': {'links_used': links, 'gb_per_s_per_link': 153.0, 'ceiling_gb_per_s': 153.0 * links,
                                    'frac_of_ceiling': (gbs / (153.0 * links)) if links else None,
                                    'note': 'bytes one rank receives / the collective\'s time, against one xGMI link per '
                                            'sending peer' + ('; ranks share one GPU here: not a link measurement' if share_gpu else '')},
                           'note': 'the collective alone, synchronous (median of 5, max over ranks)'}
    if rank == 0:
        extra['step_ms'] = ms_summary(step_ts)   # the timed steps themselves (this rank)
    if use_dist:
        # every rank's own view, and whether the asynchronous gather hides under the next step's kernels (DESIGN.md
        # section 8's open question): compute_only = the step without any collective, on this rank
        comp_ts = event_times(lambda: call(pg, feats), args.steps, warmup=2)
        mine = torch.tensor([float(np.median(step_ts)), float(np.median(comp_ts)), dt / args.steps * 1e3],
                            dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per = [[float(x) for x in t.tolist()] for t in allr]
        extra['per_rank'] = {'ms_step_with_sync_gather': [p_[0] for p_ in per], 'ms_step_compute_only': [p_[1] for p_ in per],
                             'ms_step_pipelined': [p_[2] for p_ in per],
                             'note': 'median over the timed steps of each rank, in rank order; pipelined = wall clock / steps'}
        comp, sync_, pipe = max(p_[1] for p_ in per), max(p_[0] for p_ in per), max(p_[2] for p_ in per)
        cost = sync_ - comp
        extra['gather_hidden'] = {'compute_only_ms': comp, 'with_synchronous_gather_ms': sync_, 'pipelined_ms': pipe,
                                  'gather_cost_in_a_synchronous_step_ms': cost,
                                  'hidden_fraction': (float(np.clip((sync_ - pipe) / cost, 0.0, 1.0)) if cost > 1e-3 else None),
                                  'hidden': bool(pipe <= comp * 1.03),
                                  'note': 'max over ranks; hidden = the pipelined step (gather of step i on RCCL\'s stream while '
                                          'step i + 1 synthesises, two landing buffers) costs no more than the step without any '
                                          'collective (+ 3 %)'}
    phase_done('timed_steps_and_collectives')
    if args.sustain_seconds > 0:
        sus = measure_sustained(step, args.sustain_seconds, med * 1e3, dev_index, world * B * N, sr, drain if use_dist else None)
        if use_dist:
            tt = torch.tensor([sus['ms_per_step']], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sus['ms_per_step'] = float(tt.item())
            sus['value'] = world * B * N / (sus['ms_per_step'] * 1e-3)
            sus['rtf'] = sus['value'] / sr
        if rank == 0:
            ref = float(np.median(step_ts)) if not use_dist else med * 1e3
            sus['vs_step_ms_median'] = sus['ms_per_step'] / ref
            sus['note'] = ('last-half median / median of the K synchronised steps = %.3f' % sus['vs_step_ms_median'] +
                           ('; steps include their gather (pipelined)' if use_dist else '') +
                           '; the synchronised steps pay an event wait and an idle gap each, the sustained ones run into each '
                           'other and run at whatever clock the socket power limit leaves (gpu_last_half)')
            extra['sustained'] = sus
        phase_done('sustained')
    if rank == 0 and extras_level != 'none':
        other = 'audio_only_call' if want_dict else 'outputs_dict_call'
        fn = (lambda: pg(feats)) if want_dict else (lambda: pg(feats, return_outputs_dict=True))
        ts = event_times(fn, 20, warmup=3)
        extra[other] = {'workload': 'the headline batch through ' + ('group(features): audio only' if want_dict else
                                                                      'group(features, return_outputs_dict=True)'),
                        'ms_per_step': ms_summary(ts), 'rtf': B * N / (float(np.median(ts)) * 1e-3) / sr}
        # every voice's stems (what synthesize_from_csv.py:99-120 gets by re-running processors one by one): the compacted
        # bank with the harmonic sum stopped at voice boundaries (ddspp_polyphonic_stems), per-voice noise rows, the mixer in
        # the DAG's order
        ts = event_times(lambda: pg(feats, return_outputs_dict=True, need_stems=True), 10, warmup=2)
        extra['all_stems_call'] = {'workload': 'the headline batch through group(features, return_outputs_dict=True, '
                                               'need_stems=True): the additive and noise stems of all 16 voices',
                                   'ms_per_step': ms_summary(ts), 'rtf': B * N / (float(np.median(ts)) * 1e-3) / sr}
        # ... and what its --decompose flag needs of them: the sums over the voices (ProcessorGroup.decompose: the compacted
        # bank and the noise kernel's voice sums form them anyway)
        ts = event_times(lambda: pg.decompose(feats), 10, warmup=2)
        extra['decompose_call'] = {'workload': 'the headline batch through group.decompose(features): output, dry mix, sum of '
                                               'the additive stems, sum of the noise stems (synthesize_from_csv.py:92-120)',
                                   'ms_per_step': ms_summary(ts), 'rtf': B * N / (float(np.median(ts)) * 1e-3) / sr}
        # inputs that take none of the data-dependent shortcuts of the oscillator bank: no silent voice, every
        # partial below Nyquist (low notes), every frequency moving in every frame
        fd, _ = make_features(B, P, T, H, K, S, L, device, seed=31, silent_frac=0.0, midi_lo=21, midi_hi=33, vibrato=0.004)
        pgd = build_group(dp, P, sr)
        ts = event_times(lambda: call(pgd, fd), 10, warmup=3)
        extra['dense_worst_case'] = {'workload': f'batch={B}, every voice sounding, notes A0..A1 (all {H} partials below '
                                                 'Nyquist), f0 moving in every frame (0.4 % vibrato + glide)',
                                     'ms_per_step': ms_summary(ts), 'rtf': B * N / (float(np.median(ts)) * 1e-3) / sr}
        del fd, pgd
        fm, _ = make_features(B, P, T, H, K, S, L, device, seed=32, vibrato=0.002)
        pgm = build_group(dp, P, sr)
        ts = event_times(lambda: call(pgm, fm), 10, warmup=3)
        extra['moving_f0'] = {'workload': f'the headline note mix (A0..C8, 25 % silent voices) with every f0 moving in '
                                          'every frame (0.2 % vibrato + glide)',
                              'ms_per_step': ms_summary(ts), 'rtf': B * N / (float(np.median(ts)) * 1e-3) / sr}
        del fm, pgm
        # note-shaped controls: a synthetic performance through MIDIRoll2Conditioning (onsets and releases inside the
        # segment, free voices at pitch 0 = 8.18 Hz gated by min_frequency, polyphony as it comes)
        fn_, _, st_ = make_midi_like_features(dp, B, P, T, H, K, S, L, device, seed=33)
        pgn = build_group(dp, P, sr)
        ts = event_times(lambda: call(pgn, fn_), 10, warmup=3)
        extra['midi_like'] = {'workload': f'batch={B} x {args.seconds:g} s, controls from a synthetic piano roll per segment through '
                                          'MIDIRoll2Conditioning (f0 = midi_to_hz(pitch), amplitudes re-triggered at onsets)',
                              'inputs': st_, 'ms_per_step': ms_summary(ts), 'rtf': B * N / (float(np.median(ts)) * 1e-3) / sr}
        del fn_, pgn
        f1, _ = make_features(1, P, T, H, K, S, L, device, seed=7)
        pg1 = build_group(dp, P, sr)
        d1 = min(time_steps(lambda: call(pg1, f1), 20, 3) for _ in range(3))          # best of three runs of 20
        extra['single_stream'] = {'workload': f'B=1 x {args.seconds:g} s, poly={P}', 'ms_per_segment': d1 / 20 * 1e3,
                                  'rtf': (N * 20 / d1) / sr}
        phase_done('extras_call_forms_and_inputs')
        if extras_level == 'full':
            # the same call captured once as a HIP graph and replayed (inputs copied into the captured buffers, fresh
            # noise drawn, per call): what a launch-bound caller -- one stream, streaming blocks -- would use
            from ddsp_piano_amd.graph import CapturedGroup
            cg = CapturedGroup(pg1, f1, return_outputs_dict=(args.call_form == 'outputs_dict'))
            dg = min(time_steps(lambda: cg(f1), 20, 3) for _ in range(3))
            extra['single_stream_graph'] = {'workload': extra['single_stream']['workload'] + ', hipGraph replay',
                                            'ms_per_segment': dg / 20 * 1e3, 'rtf': (N * 20 / dg) / sr}
            del cg
            # ... and the form that pays neither the Python layer nor the input copies: the one-call driver's kernels captured,
            # controls written in place into the captured buffers (CapturedGroup.inputs), fresh noise drawn per replay
            cgn = CapturedGroup(dp.NativeGroup(pg1, f1), f1, return_outputs_dict=(args.call_form == 'outputs_dict'))
            dgn = min(time_steps(lambda: cgn(), 20, 3) for _ in range(3))
            extra['single_stream_graph_native'] = {'workload': extra['single_stream']['workload'] + ', hipGraph replay of '
                                                   'ddspp_group_run, controls written in place (no per-replay input copies)',
                                                   'ms_per_segment': dgn / 20 * 1e3, 'rtf': (N * 20 / dgn) / sr}
            del cgn
            # the library's one-call driver (ddspp_group_run behind ddsp_piano_amd.NativeGroup): the same kernels enqueued
            # from C++ instead of a dozen ctypes calls -- what a caller without the Python layer gets
            ng1 = dp.NativeGroup(pg1, f1)
            dn = min(time_steps(lambda: ng1(f1, return_outputs_dict=(args.call_form == 'outputs_dict')), 20, 3) for _ in range(3))
            extra['single_stream_native'] = {'workload': extra['single_stream']['workload'] + ', ddspp_group_run',
                                             'ms_per_segment': dn / 20 * 1e3, 'rtf': (N * 20 / dn) / sr}
            del ng1
            ngb = dp.NativeGroup(pg, feats)
            ts = event_times(lambda: ngb(feats, return_outputs_dict=(args.call_form == 'outputs_dict')), 20, warmup=3)
            extra['native_group_call'] = {'workload': 'the headline batch and call form through ddspp_group_run',
                                          'ms_per_step': ms_summary(ts), 'rtf': B * N / (float(np.median(ts)) * 1e-3) / sr}
            del ngb
            phase_done('extras_full_single_stream_forms')
        # the other shipped shapes, driver-run: BASELINE config 5's per-GPU share and the dafx22 model
        for key, (b_, p_, h_, k_, s_, sr_, ir_s, note) in {
                'c5_per_gpu_share': (32, 32, 128, 96, 1, 48000, 10.0, 'BASELINE config 5 per GPU (batch 256 / 8): 48 kHz, poly 32, 10 s IR'),
                'c5_full': (256, 32, 128, 96, 1, 48000, 10.0, 'BASELINE config 5 at its STATED batch on ONE MI355X: 48 kHz, poly 32, 10 s IR (2^20-point reverb on 256 rows, 8192 voice rows)'),
                'dafx22_dims': (B, 16, 96, 64, 2, 16000, 1.5, 'configs/dafx22.gin dims: 16 kHz, two sub-strings, 1.5 s IR')}.items():
            u_ = sr_ // 250
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            fx, bx = make_features(b_, p_, T, h_, k_, s_, int(ir_s * sr_), device, seed=41)
            del bx
            pgx = build_group(dp, p_, sr_)
            ts = event_times(lambda: call(pgx, fx), 6 if key == 'c5_full' else 10, warmup=3)
            extra[key] = {'workload': f'{note}; batch={b_} x {args.seconds:g} s, H={h_}, K={k_}, S={s_}',
                          'ms_per_step': ms_summary(ts), 'value': b_ * T * u_ / (float(np.median(ts)) * 1e-3),
                          'rtf': b_ * T * u_ / (float(np.median(ts)) * 1e-3) / sr_}
            if key == 'c5_full':
                extra[key]['peak_hbm_gib'] = torch.cuda.max_memory_allocated() / 2 ** 30
                extra[key]['peak_hbm_note'] = ('torch allocator peak (inputs 15 GiB of controls + every workspace and output of the '
                                               'call) of the 288 GB on the GPU')
            if key == 'dafx22_dims':
                # the same inputs through the node list default_model.py itself builds (explicit Add nodes, noise first)
                pgd = build_default_model_group(dp, p_, sr_)
                ts = event_times(lambda: call(pgd, fx), 10, warmup=3)
                extra['default_model_dag'] = {'workload': 'ddsp_piano/default_model.py:44-80 node list at the dafx22 dims, '
                                                          f'batch={b_} x {args.seconds:g} s; with return_outputs_dict=True the '
                                                          'dictionary holds every add_i / sub_add_i as the reference\'s does '
                                                          '(round 5: every voice\'s stems are rendered for it)',
                                              'ms_per_step': ms_summary(ts),
                                              'rtf': b_ * T * u_ / (float(np.median(ts)) * 1e-3) / sr_}
                if want_dict:
                    ts = event_times(lambda: pgd(fx, return_outputs_dict=True, need_stems='last'), 10, warmup=3)
                    extra['default_model_dag_reduced_dict'] = {
                        'workload': 'the same with need_stems=\'last\' (opt-in): only the last voice\'s pair, the mix before it and '
                                    'the dry mix are written (the compacted route, round 4\'s behaviour)',
                        'ms_per_step': ms_summary(ts), 'rtf': b_ * T * u_ / (float(np.median(ts)) * 1e-3) / sr_}
                del pgd
            del fx, pgx
            torch.cuda.empty_cache()
        # every shipped gin file at its own dims and flags (dafx22.gin is `dafx22_dims` above)
        shipped = {}
        for cfg, (b_, h_, k_, s_, sr_, l_) in {
                'maestro-v2': (B, 128, 96, 1, 24000, 48000), 'dafx22-24kHz': (B, 128, 96, 2, 24000, 36000),
                'ENSTDkCl-8kHz': (B, 48, 32, 1, 8000, 16000), 'ENSTDkCl-32kHz': (B, 192, 128, 1, 32000, 64000),
                'multi_instruments': (B, 96, 64, 1, 16000, 24000), 'surrogate': (B, 96, 64, 1, 16000, 16000)}.items():
            u_ = sr_ // 250
            fx, bx = make_features(b_, 16, T, h_, k_, s_, l_, device, seed=43)
            if cfg == 'surrogate':          # per-partial decay rates and the frames since the note's onset
                gq = torch.Generator(device=device)
                gq.manual_seed(44)
                dec = 0.9990 + 0.0012 * torch.rand(b_, 16, T, h_, generator=gq, device=device)
                dt = torch.arange(T, device=device, dtype=torch.float32).view(1, 1, T, 1).expand(b_, 16, T, 1).contiguous()
                for i in range(16):
                    fx[f'decays_{i}'], fx[f'decay_time_{i}'] = dec[:, i], dt[:, i]
            pgx = build_shipped_group(dp, cfg, 16, sr_)
            ts = event_times(lambda: call(pgx, fx), 10, warmup=2)
            shipped[cfg] = {'workload': f'batch={b_} x {args.seconds:g} s, poly=16, {sr_} Hz, H={h_}, K={k_}, S={s_}, '
                                        f'IR {l_} samples' + (' (FDN node holding its parameters: their impulse response is designed at the first call and kept)'
                                                              if cfg.startswith('ENST') else '') +
                                        (' (SurrogateAdditive: per-voice rows through the fused decay kernel, batched '
                                         'route since round 4)' if cfg == 'surrogate' else ''),
                            'ms_per_step': ms_summary(ts), 'rtf': b_ * T * u_ / (float(np.median(ts)) * 1e-3) / sr_}
            if cfg.startswith('ENST'):
                # like for like with the reference, which designs the impulse response inside EVERY get_controls
                # (fdn_reverb.py:383-392): the FDN node with cache_ir=False
                pgx.processors[-1].cache_ir = False
                ts = event_times(lambda: call(pgx, fx), 10, warmup=2)
                shipped[cfg + ' (FDN designed per call)'] = {
                    'workload': shipped[cfg]['workload'].split(' (FDN node')[0] + ' (FDN impulse response designed inside every '
                                'call, as fdn_reverb.py:383-392 does: ddspp_fdn_transfer + irfft + early reflections per step)',
                    'ms_per_step': ms_summary(ts), 'rtf': b_ * T * u_ / (float(np.median(ts)) * 1e-3) / sr_}
            del fx, bx, pgx
            torch.cuda.empty_cache()
        extra['shipped_configs'] = shipped
        phase_done('extras_shipped_shapes')
        # what synthesize_midi_file.py does: the whole file as ONE segment (here 136 s, poly 16)
        Tw = 34000
        fw, _ = make_features(1, P, Tw, H, K, S, int(2.0 * sr), device, seed=11)
        pgw = build_group(dp, P, sr)
        dw = min(time_steps(lambda: call(pgw, fw), 5, 2) for _ in range(3))
        extra['whole_file'] = {'workload': f'B=1 x {Tw / 250:g} s in one segment, poly={P}, 2 s IR',
                               'ms_per_file': dw / 5 * 1e3, 'rtf': (Tw * U * 5 / dw) / sr}
        del fw, pgw
        torch.cuda.empty_cache()
        if extras_level == 'full':
            # ... and a piece of typical MAESTRO length: 20 minutes = 300 000 frames, past the 131 072 frames up to which the
            # reference's bilinear resize takes rows (t, t + 1) for every sample of frame t (core.walk_weights: the samples that
            # take row t + 1 itself are marked and the fused kernels keep the file; before round 4 it fell to per-voice
            # materialised envelopes)
            Tl = 300000
            fl, _ = make_features(1, P, Tl, H, K, S, int(2.0 * sr), device, seed=12)
            pgl = build_group(dp, P, sr)
            dl = min(time_steps(lambda: call(pgl, fl), 3, 1) for _ in range(2))
            extra['whole_file_20min'] = {'workload': f'B=1 x {Tl / 250:g} s in one segment, poly={P}, 2 s IR',
                                         'ms_per_file': dl / 3 * 1e3, 'rtf': (Tl * U * 3 / dl) / sr}
            del fl, pgl
        del f1, pg1
        torch.cuda.empty_cache()
        phase_done('extras_whole_file')
    roof_step = roof_noise = None
    if rank == 0 and not args.no_roofline:
        del feats
        torch.cuda.empty_cache()
        ceiling = measure_valu_ceiling(device)
        roof_step = measure_roofline_step(dp, base, args, T, U, device, ceiling)
        roof_noise = measure_roofline_noise(dp, base, args, T, U, device, ceiling)
        phase_done('roofline_step_and_noise')
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = measure_cpu_baseline(args, T, U, full=(extras_level == 'full'))
        phase_done('cpu_baseline')
    if dist is not None:
        dist.barrier()

    stale = [r.get('counters_stale') for r in (roof, roof_step, roof_noise) if r]
    if rank == 0:
        line = {
            'metric': 'audio samples/sec synthesized (24 kHz, poly=16), full chain; real-time factor in rtf',
            'value': value, 'unit': 'audio samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': med * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'rtf': value / sr,
            'config': {'workload': ('BASELINE config 3 per GPU: ' if (B, P, H, K, S, sr, args.seconds, L) ==
                                    (64, 16, 128, 96, 1, 24000, 3.0, 72000) else 'custom shape per GPU: ') +
                                   f'batch={B} x {args.seconds:g} s segments, '
                                   f'poly={P}, {sr} Hz, 250 Hz controls, H={H}, K={K}, S={S}, '
                                   f'{args.ir_seconds:g} s reverb IR (L={L}); global batch {world * B}'
                                   + (' = config 4' if world * B == 512 else ''),
                       'call_form': 'processor_group(features, return_outputs_dict=True) (piano_model.py:160)'
                                    if want_dict else 'processor_group(features)',
                       'global_batch': world * B, 'segment_samples': N, 'parallelism': f'batch-shard x{world}'
                                                                                       + (' + one gather of the audio per step (inside the timed step; overlapped with the next step in `pipelined`)' if world > 1 else '')},
            'roofline': roof, 'roofline_step': roof_step, 'roofline_noise': roof_noise, 'cpu_baseline': cpu,
            'counters_stale': (any(bool(x) for x in stale) if stale else None),
            'extras': extras_level, 'phase_seconds': {k: round(v, 2) for k, v in phase_s.items()},
        }
        line.update(extra)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is flushed at exit when stdout is a pipe:
        # push it out first so that the JSON line is the last line of the output
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
