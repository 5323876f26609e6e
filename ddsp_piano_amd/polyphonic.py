"""Batched execution of the polyphonic ProcessorGroup.

The reference walks 3*P + 1 DAG nodes one eager processor call at a time
(ddsp_piano/modules/polyphonic_dag.py:24-40; 4*P nodes in the older ddsp_piano/default_model.py:44-80, whose
shape is taken too since round 4).  When the DAG handed to ProcessorGroup has exactly one of those
shapes, the voices become a batch dimension: rows = B*P go through ONE get_controls kernel, ONE
fused oscillator-bank launch, ONE FIR-design + ONE time-varying-FIR launch, one mixer pass and one
rocFFT reverb.  Per-row arithmetic is that of the node-by-node walk (same kernels).  The voice sum: with every
voice's stems (need_stems=True) the mixer keeps the DAG's ((add + noise_i) + additive_i) order; the compacted
routes (audio only / outputs dict) add the voices in another order -- the oscillators of a segment are summed
across voices inside the bank -- so their mix differs from the walk's by float32 round-off of the additions
(~1e-7 relative; tests/test_gpu_group.py checks the routes against each other and against the oracle).
"""
from __future__ import annotations

import torch

from . import _lib, core
from .core import _lib_, _ptr, _stream
from .effects import FeedbackDelayNetwork, FeedbackDelayNetworkApply, Reverb
from .synths import FilteredNoise, InHarmonic, MultiAdd, SurrogateAdditive


class Plan:
    """What `recognise` found.  shape 'gin': the node list of polyphonic_dag(...) (one MultiAdd re-used for every voice);
    shape 'default_model': the node list of ddsp_piano/default_model.py:44-80 -- noise node first, explicit
    ddsp.processors.Add nodes `add_0`, then per voice `sub_add_i` (noise + additive) and `add_i` (add_{i-1} + sub_add_i),
    the reverb fed from `add_{P-1}`: `adds` / `subs` hold those processors per voice (subs[0] is None)."""

    def __init__(self, additive, noise, add, reverb, additive_keys, noise_keys, reverb_keys, n_synths,
                 shape='gin', adds=None, subs=None):
        self.additive, self.noise, self.add, self.reverb = additive, noise, add, reverb
        self.additive_keys, self.noise_keys, self.reverb_keys = additive_keys, noise_keys, reverb_keys
        self.n_synths = n_synths
        self.shape, self.adds, self.subs = shape, adds, subs


def _reverb_node(node, feed):
    """(reverb, reverb_keys) of a last node fed from `feed`, or None when it is not one the batched route takes."""
    reverb = node[0]
    # ddsp.effects.Reverb with the impulse response as a control (maestro-v2.gin:152-153), or holding its own
    # (trainable=True); a FeedbackDelayNetwork that holds its parameters, reverb_controls = []
    # (ENSTDkCl-8kHz.gin:85-86) or takes them as controls; the FDN apply step
    if not isinstance(reverb, (Reverb, FeedbackDelayNetworkApply, FeedbackDelayNetwork)):
        return None
    if not node[1] or node[1][0] != feed:
        return None
    reverb_keys = list(node[1][1:])
    if any('/' in k for k in reverb_keys):
        return None
    return reverb, reverb_keys


def _plain_keys(keys, n):
    return len(keys) == n and not any('/' in k for k in keys)


def recognise(dag):
    """Return a Plan if ``dag`` is polyphonic_dag(...)'s node list, or default_model.py's, over this package's processors."""
    return _recognise_gin(dag) or _recognise_default_model(dag)


def _recognise_gin(dag):
    n = len(dag)
    if n < 3:
        return None
    has_reverb = (n % 3) == 1
    if n % 3 not in (0, 1):
        return None
    p = n // 3
    additive, noise, add = dag[0][0], dag[1][0], dag[2][0]
    # (SurrogateAdditive, configs/surrogate.gin: six controls -- amplitudes, decays, decay_time, harmonic_distribution,
    # inharm_coef, f0_hz -- and per-voice rows through the fused decay kernel; round 4)
    if not (isinstance(additive, (InHarmonic, SurrogateAdditive)) and isinstance(noise, FilteredNoise)
            and isinstance(add, MultiAdd)):
        return None
    n_add = 6 if isinstance(additive, SurrogateAdditive) else 4
    additive_keys, noise_keys = [], []
    for i in range(p):
        a, z, m = dag[3 * i], dag[3 * i + 1], dag[3 * i + 2]
        if a[0] is not additive or z[0] is not noise or m[0] is not add:
            return None
        if not _plain_keys(a[1], n_add) or not _plain_keys(z[1], 1):
            return None
        expect = [noise.name + '/signal', additive.name + '/signal']
        if i > 0:
            expect = [add.name + '/signal'] + expect
        if list(m[1]) != expect:
            return None
        additive_keys.append(list(a[1]))
        noise_keys.append(z[1][0])
    reverb, reverb_keys = None, []
    if has_reverb:
        found = _reverb_node(dag[-1], add.name + '/signal')
        if found is None:
            return None
        reverb, reverb_keys = found
    return Plan(additive, noise, add, reverb, additive_keys, noise_keys, reverb_keys, p)


def _recognise_default_model(dag):
    """ddsp_piano/default_model.py:44-80:  noise(magnitudes_0), additive(.._0), add_0(noise/signal, additive/signal);
    per further voice: additive(.._i), noise(magnitudes_i), sub_add_i(noise/signal, additive/signal),
    add_i(add_{i-1}/signal, sub_add_i/signal); then the reverb on add_{P-1}/signal (3 + 4 (P - 1) [+ 1] nodes)."""
    from .processors import Add
    n = len(dag)
    if n < 3 or (n - 3) % 4 not in (0, 1):
        return None
    has_reverb = (n - 3) % 4 == 1
    p = 1 + (n - 3) // 4
    noise, additive, add0 = dag[0][0], dag[1][0], dag[2][0]
    if not (isinstance(additive, InHarmonic) and isinstance(noise, FilteredNoise) and type(add0) is Add):
        return None
    if not _plain_keys(dag[0][1], 1) or not _plain_keys(dag[1][1], 4):
        return None
    pair = [noise.name + '/signal', additive.name + '/signal']
    if list(dag[2][1]) != pair:
        return None
    additive_keys, noise_keys = [list(dag[1][1])], [dag[0][1][0]]
    adds, subs = [add0], [None]
    for i in range(1, p):
        a, z, sb, m = dag[4 * i - 1], dag[4 * i], dag[4 * i + 1], dag[4 * i + 2]
        if a[0] is not additive or z[0] is not noise or type(sb[0]) is not Add or type(m[0]) is not Add:
            return None
        if not _plain_keys(a[1], 4) or not _plain_keys(z[1], 1):
            return None
        if list(sb[1]) != pair or list(m[1]) != [adds[-1].name + '/signal', sb[0].name + '/signal']:
            return None
        additive_keys.append(list(a[1]))
        noise_keys.append(z[1][0])
        adds.append(m[0])
        subs.append(sb[0])
    # every Add node is an object and a name of its own (a re-used one would be overwritten in the outputs dictionary)
    objs = adds + subs[1:]
    names = [o.name for o in objs] + [noise.name, additive.name]
    if len({id(o) for o in objs}) != len(objs) or len(set(names)) != len(names):
        return None
    reverb, reverb_keys = None, []
    if has_reverb:
        found = _reverb_node(dag[-1], adds[-1].name + '/signal')
        if found is None or found[0].name in names:
            return None
        reverb, reverb_keys = found
    return Plan(additive, noise, adds[-1], reverb, additive_keys, noise_keys, reverb_keys, p,
                shape='default_model', adds=adds, subs=subs)


def pick_voice_sums(n_segments, n_voices, n_frames, forced=0):
    """Voices summed per row inside the FilteredNoise kernel (8, 4, 2 or 1 = per-voice rows): the largest divisor of the voice
    count that still leaves the kernel 768 units of (row, 30-frame window) -- the workgroups the chip holds; `forced` > 0:
    the largest divisor up to it, whatever the batch size (DDSPP_VOICE_SUMS).  csrc/group.cpp applies the same rule."""
    for v in (8, 4, 2):
        if n_voices % v == 0 and (v <= forced if forced > 0 else
                                  n_segments * (n_voices // v) * -(-n_frames // 30) >= 768):
            return v
    return 1


_side_streams = {}


def _side_stream(device):
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)     # one per caller stream
    st = _side_streams.get(key)
    if st is None:
        st = _side_streams[key] = torch.cuda.Stream(device=device)
    return st


def _same_buffer_slices(tensors, step_elems):
    """True when tensors[i] starts i * step_elems elements after tensors[0] inside one storage (the storage is asked of
    the first and the last only: slices at exact multiples in between lie inside it)."""
    t0, tl = tensors[0], tensors[-1]
    if t0.untyped_storage().data_ptr() != tl.untyped_storage().data_ptr():
        return False
    step, ptr = step_elems * t0.element_size(), t0.data_ptr()
    for x in tensors:
        if x.data_ptr() != ptr:
            return False
        ptr += step
    return True


def _stack_voices(tensors, voice_major=None):
    """The P per-voice tensors [B, T, C] as ONE buffer of B * P rows, plus its row order.

    Returns (rows [B * P, T, C] contiguous, voice_major).  Zero-copy when the P tensors are the P slices of
    one buffer, in either of the two orders a batched control network produces:
      * voice major [P, B, T, C]: what the reference's Parallelizer.unparallelize hands over
        (`features[k + f'_{i}'] = features[k][i]`, sub_modules.py:586-592);
      * segment major [B, P, T, C].
    Otherwise the tensors are copied into the requested order (default: voice major, P block copies)."""
    t0 = tensors[0]
    p = len(tensors)
    shape0, stride0 = t0.shape, t0.stride()

    def alike():
        for x in tensors:
            if x.shape != shape0 or x.stride() != stride0 or x.dtype != torch.float32:
                return False
        return True
    if t0.is_cuda and t0.dtype == torch.float32 and t0.dim() == 3 and alike():
        b, t, c = shape0
        sb, st, sc = stride0
        if sc == 1 and st == c:
            if voice_major in (None, True) and sb == t * c and _same_buffer_slices(tensors, b * t * c):
                return torch.as_strided(t0, (p * b, t, c), (t * c, c, 1)), True
            if voice_major in (None, False) and sb == p * t * c and _same_buffer_slices(tensors, t * c):
                return torch.as_strided(t0, (b * p, t, c), (t * c, c, 1)), False
    vm = True if voice_major is None else voice_major
    xs = [core.tf_float32(x) for x in tensors]
    b, t, c = xs[0].shape
    return torch.stack(xs, dim=0 if vm else 1).reshape(b * p, t, c), vm


def noise_rows(noise, B, P, N, voice_major):
    """The caller's explicit noise -- [B, P, N], or a sequence of P tensors [B, N] (voice i = the i-th call of the
    noise processor in the DAG) -- as the [R, N] rows the batched kernels take, in the row order of the controls."""
    if isinstance(noise, (list, tuple)):
        if len(noise) != P:
            raise ValueError(f'noise: expected {P} per-voice tensors, got {len(noise)}')
        rows = torch.stack([core.tf_float32(z) for z in noise], dim=0 if voice_major else 1)
    else:
        rows = core.tf_float32(noise)
        if rows.dim() == 2 and P == 1:
            rows = rows[:, None, :]
        if tuple(rows.shape) != (B, P, N):
            raise ValueError(f'noise must be [batch, n_synths, n_samples] = {(B, P, N)}, got {tuple(rows.shape)}')
        if voice_major:
            rows = rows.transpose(0, 1)
    if tuple(rows.shape) != ((P, B, N) if voice_major else (B, P, N)):
        raise ValueError(f'noise: every per-voice tensor must be {(B, N)}')
    return rows.contiguous().reshape(B * P, N)


def run(plan, inputs, noise=None, need_stems=True):
    """Execute the polyphonic DAG with voices batched.  Returns the ddsp-style outputs dict, or None
    when the inputs do not fit the batched kernels (caller then walks the DAG node by node).

    noise: the uniform(-1, 1) draws of the noise processor given explicitly ([B, P, N] or P tensors [B, N]; the
    reference draws them unseeded, filtered_noise_synth.py:39-40); None = the processor's own Philox draw.

    need_stems=False (a plain ``group(features)`` call that only wants the audio): the additive branch runs through
    the compacted kernel, which forms the per-segment mix directly -- the `additive` / `noise` / `voices` entries
    are then absent from the outputs dict.
    need_stems='last' (``group(features, return_outputs_dict=True)``, what PianoModel.call does,
    piano_model.py:160): the same fast mix, plus the entries the reference's dict has -- the re-used processors'
    outputs, i.e. the LAST voice's stems and controls.
    need_stems=True / 'all': every voice's stems ([B, P, N] under outputs['voices']), per-voice kernels.
    need_stems='sums': the audio-only route plus outputs['voices_sum'] = {'additive': sum over the voices of the additive
    stems, 'noise': sum of the noise stems} -- what synthesize_from_csv.py:99-120 (--decompose) adds up by calling the two
    processors once per voice; the compacted bank and the noise kernel's voice sums form them anyway."""
    P = plan.n_synths
    default_shape = plan.shape == 'default_model'
    surrogate = isinstance(plan.additive, SurrogateAdditive)
    add_ctl = [[inputs[k[j]] for k in plan.additive_keys] for j in range(6 if surrogate else 4)]
    if surrogate:                                    # (amplitudes, decays, decay_time, harmonic_distribution, inharm_coef, f0_hz)
        dec_in, dtm_in = add_ctl[1], add_ctl[2]
        add_ctl = [add_ctl[0], add_ctl[3], add_ctl[4], add_ctl[5]]
        if any(x is None for x in dec_in + dtm_in):
            return None
    hd, vm = _stack_voices(add_ctl[1])               # [R, T, H]; the widest control decides the row order
    amp, _ = _stack_voices(add_ctl[0], vm)           # [R, T, 1]
    inh, _ = _stack_voices(add_ctl[2], vm)           # [R, T, 1]
    f0, _ = _stack_voices(add_ctl[3], vm)            # [R, T, S]
    if surrogate:
        dec, _ = _stack_voices(dec_in, vm)           # [R, T, H]
        dtm, _ = _stack_voices(dtm_in, vm)           # [R, T, 1]
    mags, _ = _stack_voices([inputs[k] for k in plan.noise_keys], vm)    # [R, T, K]
    R, T, H = hd.shape
    B = R // P
    S = f0.shape[-1]
    K = mags.shape[-1]
    if amp.shape[-1] != 1 or inh.shape[-1] != 1 or mags.shape[1] != T:
        return None
    additive, noise_p = plan.additive, plan.noise
    if not isinstance(additive, (InHarmonic, SurrogateAdditive)):
        return None
    if surrogate and (S != 1 or tuple(dec.shape) != (R, T, H) or tuple(dtm.shape) != (R, T, 1) or not additive.inference):
        return None
    U = int(additive.sample_rate / additive.frame_rate) if surrogate else additive.upsampling
    N = U * T
    if not core.fused_synthesis_supported(T, N):
        return None
    if noise_p._n_samples(mags) != N:
        return None
    if S != 1 and type(additive) is InHarmonic:
        return None
    dev = hd.device
    vmi = 1 if vm else 0

    want_all = need_stems is True or need_stems == 'all'
    want_last = need_stems == 'last'
    compact = (not want_all) and additive.inference and P * S <= 64 and N % 4 == 0
    # every voice's stems (round 5, late): the compacted bank as well -- core.polyphonic_stems, the group route's packing with
    # the harmonic sum stopped at voice boundaries (lanes only for audible partials, silent rows cost nothing, the memoised
    # pre-pass and the compacted scan as on the group route); more than 64 (voice, sub-string) rows per segment: every voice
    # as a segment of its own.  Same box, 1024 rows of config 3, per-voice fused kernel -> one-voice segments -> packed:
    # 2.03 -> 1.84 -> 1.19 ms; two sub-strings 4.66 -> 2.49 -> 2.14; every f0 moving 3.81 -> 3.45 -> 2.15.
    stems_compact = (want_all and additive.inference and S <= 64 and S * H <= 512 and N % 4 == 0 and
                     not _lib.options.no_stems_compact)
    if surrogate:                  # (the compacted bank takes the decay term when get_controls runs as kernels)
        compact = compact and core.scale_kind(additive.scale_fn) is not None and H <= 512
        stems_compact = False
    # --- noise branch ---------------------------------------------------------------------------
    fuse_scale = (compact or stems_compact) and noise_p.scale_fn is not None and noise_p.raw_scale() is not None
    # audio only: the noise kernel adds the filtered noise of up to 8 voices of a segment in registers
    # (batch 64, same box: 2.12 ms per step with per-voice rows, 2.08 / 2.05 / 2.03 / 2.04 with 2 / 4 / 8 / 16)
    opt = _lib.options
    # ... when the rows alone give the kernel workgroups enough: a window is 30 frames, the chip holds 768 workgroups -- a
    # single 3 s segment is 50 units of eight voices, or 400 of one (0.25 -> 0.19 ms for the segment)
    voice_sums = pick_voice_sums(B, P, T, opt.voice_sums)
    if not (compact and voice_sums > 1 and P % voice_sums == 0 and
            not opt.no_voice_sums):
        voice_sums = 1

    last_stem = {}

    def noise_branch(noise):
        nctl = None if fuse_scale else noise_p.get_controls(mags)    # audio only: scale_fn runs inside the FIR design
        if noise is None:
            noise = noise_p.draw_noise_lazy(R, N, dev)    # (round 6) drawn inside the filter kernel when its shape allows
        else:
            noise = noise_rows(noise, B, P, N, vm)
        m_src = mags if fuse_scale else nctl['magnitudes']
        rs = noise_p.raw_scale() if fuse_scale else None
        if voice_sums > 1:
            # the filtered noise of several voices leaves the kernel as one row (an eighth of the round trip); for the
            # outputs dictionary the last voice stays out of its row's sum and leaves on its own
            res = core.frequency_filter_voice_sums(noise, m_src, noise_p.window_size, rs, P, voice_sums, vm,
                                                   split_last=want_last)
            if res is not None:
                if want_last:
                    res, last_stem['noise'] = res
                return nctl, res, voice_sums
        sig = core.frequency_filter(noise, m_src, window_size=noise_p.window_size, raw_scale=rs)
        return nctl, sig, 1

    # The noise branch does not depend on the additive one until the mix: it is enqueued on a side stream first, so
    # the latency-bound parts of the additive chain (the one-wavefront-per-row pre-pass, kernel tails) overlap with it.
    # (worth the two stream joins only for large batches / long files: 2 % there, a loss for a single 3 s segment)
    side = _side_stream(dev) if (dev.type == 'cuda' and R * N >= opt.side_stream_min and not opt.no_side_stream and
                                 not torch.cuda.is_current_stream_capturing()) else None
    rev_state = None
    if side is not None:
        cur = torch.cuda.current_stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            if type(plan.reverb) is Reverb and len(plan.reverb_keys) == 1 and not opt.no_early_ir:
                # the room's impulse response is an input: its spectrum is ready long before the dry mix
                rev_state = plan.reverb.begin(B, N, inputs[plan.reverb_keys[0]], key_stream=cur.cuda_stream)
            nctl, noise_sig, noise_vq = noise_branch(noise)
    else:
        nctl, noise_sig, noise_vq = noise_branch(noise)
    # --- additive branch ------------------------------------------------------------------------
    # compacted route: harmonic_shifts are never written (the bank forms them from inharm_coef per lane and frame)
    if surrogate:
        # SurrogateAdditive: its get_controls over all rows at once, per-voice rows straight from the frame controls with
        # the decay term inside the oscillator kernel (core.surrogate_harmonic_synthesis -> ddspp_surrogate_harmonic_synthesis)
        ctl = additive.get_controls(amp, dec, dtm, hd, inh, f0, _want_counts=compact)
        if compact and '_audible' not in ctl:          # get_controls did not run as kernels: the caller walks the DAG
            return None
    else:
        lean = compact or stems_compact
        # (round 6) on both compacted routes the bank is the only reader of the normalised distribution: the rows are
        # written only below each frame's audible count (the last voice's whole, for the outputs dictionary)
        to_bank = compact or (stems_compact and P * S <= 64 and not _lib.options.stems_single)
        ctl = additive._controls(amp, hd, inh, f0, want_counts=lean, want_shifts=not lean,
                                 last_voice_of=(P, vm) if (want_last or stems_compact) else None,
                                 sparse_for_bank=(P, vm) if to_bank else None)
    additive_last = None
    if surrogate and compact:
        additive_mix = core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T),
                                                ctl['harmonic_distribution'], None, B, N,
                                                additive.sample_rate, voice_major=vm, audible=ctl['_audible'],
                                                split_last=want_last, inharm_coef=ctl['_inharm_coef'].reshape(R, T),
                                                decays=ctl['decays'],
                                                decay_time=core.tf_float32(ctl['decay_time']).reshape(R, T).contiguous())
        if want_last:
            additive_mix, additive_last = additive_mix
        additive_sig = None
    elif surrogate:
        additive_sig = additive.get_signal(**{k: v for k, v in ctl.items() if not k.startswith('_')})
    elif compact:
        additive_mix = core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T),
                                                ctl['harmonic_distribution'], None, B, N,
                                                additive.sample_rate, voice_major=vm, audible=ctl['_audible'],
                                                split_last=want_last, inharm_coef=ctl['_inharm_coef'].reshape(R, T))
        if want_last:               # (voices 0 .. P-2 summed, the last voice's stem): one launch, no oscillator twice
            additive_mix, additive_last = additive_mix
        additive_sig = None
    elif stems_compact and P * S <= 64 and not _lib.options.stems_single:
        # the voices of a segment packed into the same wavefronts, the harmonic sum stopped at voice boundaries
        additive_sig = core.polyphonic_stems(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T), ctl['harmonic_distribution'],
                                             None, B, N, additive.sample_rate, voice_major=vm, audible=ctl['_audible'],
                                             inharm_coef=ctl['_inharm_coef'].reshape(R, T))
    elif stems_compact:
        # every voice a segment of its own (more than 64 (voice, sub-string) rows per segment; DDSPP_STEMS_SINGLE=1)
        additive_sig = core.polyphonic_additive(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T), ctl['harmonic_distribution'],
                                                None, R, N, additive.sample_rate, audible=ctl['_audible'],
                                                inharm_coef=ctl['_inharm_coef'].reshape(R, T))
    else:
        additive_sig = core.harmonic_synthesis_fused(ctl['f0_hz'], ctl['amplitudes'].reshape(R, T),
                                                     ctl['harmonic_distribution'], ctl['harmonic_shifts'], N,
                                                     additive.sample_rate, additive.inference)
    if side is not None:
        cur.wait_stream(side)
        noise_sig.record_stream(cur)
        if nctl is not None:
            nctl['magnitudes'].record_stream(cur)
        if 'noise' in last_stem:
            last_stem['noise'].record_stream(cur)
        if rev_state is not None:
            rev_state['ws'].record_stream(cur)

    def run_reverb(dry):
        if rev_state is not None:
            ir = inputs[plan.reverb_keys[0]]
            return {'signal': plan.reverb.finish(rev_state, dry), 'controls': {'audio': dry, 'ir': ir}}
        return plan.reverb(dry, *[inputs[k] for k in plan.reverb_keys], return_outputs_dict=True)

    def per_voice(x, shape):       # rows -> [B, P, ...] (a transposed view when the rows are voice major)
        return x.reshape((P, B) + shape).transpose(0, 1) if vm else x.reshape((B, P) + shape)

    last = P - 1

    def voice(x, shape):
        return per_voice(x, shape)[:, last]

    # --- add chain ------------------------------------------------------------------------------
    dry = torch.empty((B, N), dtype=torch.float32, device=dev)
    if compact:
        outputs = {'inputs': inputs}
        outputs.update(inputs)
        add_controls = {}
        pz = P // noise_vq if noise_vq > 1 else P          # noise rows per segment ([B, P / noise_vq] sums are segment major)
        zvm = 0 if noise_vq > 1 else vmi
        if want_last:
            # What the node-by-node walk leaves behind: the three re-used processors hold the LAST voice's stems and
            # controls (polyphonic_dag.py re-uses the objects), and the last `add` call's operands are (the mix of
            # the voices before it, its noise, its additive signal).  The bank and the noise kernel kept the last
            # voice apart, so the chain's last step is evaluated as the DAG writes it: (prev + noise) + additive.
            noise_last = last_stem.get('noise')
            if noise_last is None:                          # per-voice noise rows: the last voice is a row block
                noise_last = voice(noise_sig, (N,)).contiguous()
                z_rows, z_n = per_voice(noise_sig, (N,))[:, :last].contiguous(), P - 1
                zvm = 0
            else:
                z_rows, z_n = noise_sig, pz
            prev = torch.empty((B, N), dtype=torch.float32, device=dev)
            sub = None
            if P > 1 and default_shape:
                # default_model.py:68-74: the last voice is added as a pair, add_{P-1} = add_{P-2} + (noise + additive)
                sub = torch.empty((B, N), dtype=torch.float32, device=dev)
                _lib.check(_lib_().ddspp_mix_last_voice_paired(_ptr(additive_mix), 1, _ptr(z_rows), z_n, _ptr(noise_last),
                                                               _ptr(additive_last), _ptr(prev), _ptr(sub), _ptr(dry), B, N,
                                                               zvm, _stream()))
            elif P > 1:
                _lib.check(_lib_().ddspp_mix_last_voice(_ptr(additive_mix), 1, _ptr(z_rows), z_n, _ptr(noise_last),
                                                        _ptr(additive_last), _ptr(prev), _ptr(dry), B, N, zvm, _stream()))
            else:
                dry = core.add_signals([noise_last, additive_last])
            lc = {k: voice(ctl[k], sh) for k, sh in (('amplitudes', (T, 1)), ('harmonic_distribution', (T, H)),
                                                     ('f0_hz', (T, S)))}
            if surrogate:
                lc.update(harmonic_shifts=voice(ctl['harmonic_shifts'], (T, H)), decays=voice(ctl['decays'], (T, H)),
                          decay_time=voice(ctl['decay_time'], (T, 1)))
            else:
                lc['harmonic_shifts'] = ctl['_shifts_last']   # written by the get_controls kernel for the last voice only
            mags_last = voice(nctl['magnitudes'], (T, K)) if nctl is not None else \
                noise_p.get_controls(voice(mags, (T, K)).contiguous())['magnitudes']
            outputs[additive.name] = {'signal': additive_last, 'controls': lc}
            outputs[noise_p.name] = {'signal': noise_last, 'controls': {'magnitudes': mags_last}}
            if default_shape:
                # What this route forms of the dictionary default_model.py's node list leaves: the last voice's pair
                # (`sub_add_{P-1}`), the mix before it (`add_{P-2}`, signal only) and the dry mix (`add_{P-1}`).  The
                # `add_i` / `sub_add_i` of the voices before are running sums of stems the compacted bank never
                # forms: need_stems=True renders them all.
                if P > 1:
                    outputs[plan.subs[last].name] = {'signal': sub,
                                                     'controls': {'signal_one': noise_last, 'signal_two': additive_last}}
                    outputs[plan.adds[last - 1].name] = {'signal': prev, 'controls': {}}
                    add_controls = {'signal_one': prev, 'signal_two': sub}
                else:
                    add_controls = {'signal_one': noise_last, 'signal_two': additive_last}
            else:
                add_controls = {'signal_0': prev, 'signal_1': noise_last, 'signal_2': additive_last} if P > 1 else \
                    {'signal_0': noise_last, 'signal_1': additive_last}
        else:
            _lib.check(_lib_().ddspp_mix_voices(_ptr(additive_mix), 1, _ptr(noise_sig), pz, _ptr(dry), B, N, N, zvm,
                                                _stream()))
            if need_stems == 'sums':
                noise_sum = noise_sig.reshape(pz, B, N).sum(0) if zvm else noise_sig.reshape(B, pz, N).sum(1)
                outputs['voices_sum'] = {'additive': additive_mix, 'noise': noise_sum}
        outputs[plan.add.name] = {'signal': dry, 'controls': add_controls}
        module_outputs = outputs[plan.add.name]
        if plan.reverb is not None:
            module_outputs = run_reverb(dry)
            outputs[plan.reverb.name] = module_outputs
        outputs['out'] = module_outputs
        return outputs
    outputs = {'inputs': inputs}
    outputs.update(inputs)
    if default_shape:
        # every `sub_add_i` and `add_i` of default_model.py:56-74, in the DAG's order: add_0 = noise_0 + additive_0,
        # sub_add_i = noise_i + additive_i, add_i = add_{i-1} + sub_add_i
        # (one pass over the stems, ddspp_add_chain_paired, instead of 2 P - 1 add kernels on strided views: 1.1 -> 0.25 ms
        # at config 3)
        subs = torch.empty((B, P, N), dtype=torch.float32, device=dev)
        runs = torch.empty((B, P, N), dtype=torch.float32, device=dev)
        _lib.check(_lib_().ddspp_add_chain_paired(_ptr(additive_sig), _ptr(noise_sig), _ptr(subs), _ptr(runs), B, P, N, vmi,
                                                  _stream()))
        additive_sig = per_voice(additive_sig, (N,))
        noise_sig = per_voice(noise_sig, (N,))
        outputs[plan.adds[0].name] = {'signal': runs[:, 0],
                                      'controls': {'signal_one': noise_sig[:, 0], 'signal_two': additive_sig[:, 0]}}
        for i in range(1, P):
            outputs[plan.subs[i].name] = {'signal': subs[:, i],
                                          'controls': {'signal_one': noise_sig[:, i], 'signal_two': additive_sig[:, i]}}
            outputs[plan.adds[i].name] = {'signal': runs[:, i], 'controls': {'signal_one': runs[:, i - 1], 'signal_two': subs[:, i]}}
        dry = runs[:, P - 1].contiguous()                 # (the reverb takes rows N apart)
        outputs[plan.adds[P - 1].name]['signal'] = dry
    else:
        prev = torch.empty((B, N), dtype=torch.float32, device=dev) if P > 1 else None
        _lib.check(_lib_().ddspp_polyphonic_mix(_ptr(additive_sig), _ptr(noise_sig), _ptr(dry), _ptr(prev), B, P, N, N,
                                                vmi, _stream()))
        additive_sig = per_voice(additive_sig, (N,))
        noise_sig = per_voice(noise_sig, (N,))
    outputs[additive.name] = {
        'signal': additive_sig[:, last],
        'controls': {'amplitudes': voice(ctl['amplitudes'], (T, 1)),
                     'harmonic_distribution': voice(ctl['harmonic_distribution'], (T, H)),
                     'harmonic_shifts': ctl['_shifts_last'] if stems_compact else voice(ctl['harmonic_shifts'], (T, H)),
                     'f0_hz': voice(ctl['f0_hz'], (T, S))}}
    if surrogate:
        outputs[additive.name]['controls'].update(decays=voice(ctl['decays'], (T, H)),
                                                  decay_time=voice(ctl['decay_time'], (T, 1)))
    mags_last = voice(nctl['magnitudes'], (T, K)) if nctl is not None else \
        noise_p.get_controls(voice(mags, (T, K)).contiguous())['magnitudes']
    outputs[noise_p.name] = {'signal': noise_sig[:, last], 'controls': {'magnitudes': mags_last}}
    if not default_shape:
        add_controls = {'signal_0': prev, 'signal_1': noise_sig[:, last], 'signal_2': additive_sig[:, last]} if P > 1 else \
            {'signal_0': noise_sig[:, last], 'signal_1': additive_sig[:, last]}
        outputs[plan.add.name] = {'signal': dry, 'controls': add_controls}
    module_outputs = outputs[plan.add.name]
    # every voice's stems, which the reference can only get by re-running processors one by one
    # (synthesize_from_csv.py:99-120)
    outputs['voices'] = {'additive': additive_sig, 'noise': noise_sig}

    if plan.reverb is not None:
        module_outputs = run_reverb(dry)
        outputs[plan.reverb.name] = module_outputs
    outputs['out'] = module_outputs
    return outputs
