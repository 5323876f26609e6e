"""GPU parity: FilteredNoise (FIR design + time-varying FIR) and the rocFFT reverb vs the oracle."""
import numpy as np
import pytest
import torch

from util import O, rms, rms_err, set_option, synth_ir

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _dev(x):
    return torch.as_tensor(x, device='cuda')


@pytest.mark.parametrize('K,window_size', [(96, 257), (64, 257), (32, 257), (128, 257), (200, 257), (65, 0),
                                           (129, 257)])
def test_frequency_impulse_response(K, window_size):
    from ddsp_piano_amd import core
    rng = np.random.default_rng(K)
    mags = rng.uniform(0, 2, [2, 13, K]).astype(np.float32)
    ref = O.frequency_impulse_response(mags, window_size)
    got = core.frequency_impulse_response(_dev(mags), window_size).cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 5e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('B,T,U,K', [(2, 40, 96, 96), (2, 30, 64, 64), (3, 50, 32, 32), (1, 20, 128, 128),
                                     (2, 12, 192, 96), (2, 750, 96, 96), (1, 9, 100, 20)])
def test_frequency_filter_matches_oracle(B, T, U, K):
    from ddsp_piano_amd import core
    rng = np.random.default_rng(T + U)
    mags = rng.uniform(0, 1, [B, T, K]).astype(np.float32) ** 4
    noise = rng.uniform(-1, 1, [B, T * U]).astype(np.float32)
    ref = O.frequency_filter(noise, mags, window_size=257)
    got = core.frequency_filter(_dev(noise), _dev(mags), window_size=257).cpu().numpy()
    err = rms_err(got, ref)
    assert err < TOL * max(1.0, rms(ref)), f'{err:.3e} vs rms {rms(ref):.3e}'


def test_tiled_and_generic_fir_agree(monkeypatch):
    from ddsp_piano_amd import core
    rng = np.random.default_rng(2)
    B, T, U, K = 2, 60, 96, 96
    mags = _dev(rng.uniform(0, 1, [B, T, K]).astype(np.float32))
    noise = _dev(rng.uniform(-1, 1, [B, T * U]).astype(np.float32))
    a = core.frequency_filter(noise, mags, window_size=257)
    set_option(monkeypatch, 'DDSPP_FIR_GENERIC', '1')
    b = core.frequency_filter(noise, mags, window_size=257)
    assert (a - b).abs().max().item() < 2e-6


def test_flat_magnitudes_delay_noise_by_two_samples():
    """KAT: a flat spectrum yields the input delayed by 2 samples, gain 1."""
    from ddsp_piano_amd import core
    rng = np.random.default_rng(0)
    noise = rng.uniform(-1, 1, [1, 40 * 96]).astype(np.float32)
    got = core.frequency_filter(_dev(noise), torch.ones(1, 40, 96, device='cuda'), 257).cpu().numpy()
    assert np.abs(got[0, 2:] - noise[0, :-2]).max() < 2e-6


def test_filtered_noise_processor():
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(6)
    B, T, K, sr = 2, 30, 96, 24000
    raw = rng.normal(0, 1, [B, T, K]).astype(np.float32)
    noise = rng.uniform(-1, 1, [B, T * 96]).astype(np.float32)
    o = O.FilteredNoise(frame_rate=250, sample_rate=sr)
    g = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=sr)
    octl = o.get_controls(raw)
    gctl = g.get_controls(_dev(raw))
    np.testing.assert_allclose(gctl['magnitudes'].cpu().numpy(), octl['magnitudes'], rtol=2e-5, atol=1e-9)
    ref = o.get_signal(octl['magnitudes'], noise=noise)
    got = g.get_signal(_dev(octl['magnitudes']), noise=_dev(noise)).cpu().numpy()
    assert rms_err(got, ref) < TOL
    # unseeded draw: right shape, right range, different on every call
    a, b = g(_dev(raw)), g(_dev(raw))
    assert a.shape == (B, T * 96) and not torch.equal(a, b)


def test_uniform_noise_statistics():
    from ddsp_piano_amd import core
    x = core.uniform_noise((4, 100000), seed=3).cpu().numpy()
    assert x.min() >= -1.0 and x.max() < 1.0
    assert abs(x.mean()) < 5e-3 and abs(x.std() - 1 / np.sqrt(3)) < 5e-3
    y = core.uniform_noise((4, 100000), seed=3).cpu().numpy()
    z = core.uniform_noise((4, 100000), seed=4).cpu().numpy()
    assert np.array_equal(x, y) and not np.array_equal(x, z)


@pytest.mark.parametrize('B,N,L,add_dry', [(2, 7200, 4800, True), (3, 24000, 24000, True),
                                           (2, 72000, 48000, False), (1, 72000, 72000, True)])
def test_reverb_matches_oracle(B, N, L, add_dry):
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(N + L)
    audio = rng.normal(0, 0.1, [B, N]).astype(np.float32)
    ir = synth_ir(rng, B, L)
    ref = O.Reverb(add_dry=add_dry).get_signal(audio, ir)
    got = dp.Reverb(add_dry=add_dry).get_signal(_dev(audio), _dev(ir)).cpu().numpy()
    err = rms_err(got, ref)
    assert err < TOL * max(1.0, rms(ref)), f'{err:.3e} vs rms {rms(ref):.3e}'


def test_reverb_kats():
    import ddsp_piano_amd as dp
    rng = np.random.default_rng(1)
    audio = rng.normal(0, 1, [2, 4096]).astype(np.float32)
    d = 37
    ir = np.zeros([2, 512], np.float32)
    ir[:, d] = 1.0
    ir[:, 0] = 5.0                                  # the dry tap is masked away
    got = dp.Reverb().get_signal(_dev(audio), _dev(ir)).cpu().numpy()
    ref = audio.copy()
    ref[:, d:] += audio[:, :-d]
    assert np.abs(got - ref).max() < 1e-5
    # ir with only the masked tap: out = dry
    ir0 = np.zeros([2, 512], np.float32)
    ir0[:, 0] = 1.0
    assert np.abs(dp.Reverb().get_signal(_dev(audio), _dev(ir0)).cpu().numpy() - audio).max() < 1e-5
    # shared 1-D impulse response, 3-D [B, L, 1] impulse response
    one = dp.Reverb().get_signal(_dev(audio), _dev(ir[0])).cpu().numpy()
    three = dp.Reverb().get_signal(_dev(audio), _dev(ir[:, :, None])).cpu().numpy()
    assert np.abs(one - ref).max() < 1e-5 and np.abs(three - ref).max() < 1e-5
    with pytest.raises(ValueError):
        dp.Reverb().get_controls(_dev(audio))        # non-trainable reverb needs an ir
    with pytest.raises(ValueError):
        dp.Reverb().get_signal(_dev(audio), _dev(np.zeros([3, 512], np.float32)))


def test_fdn_apply_and_fft_convolve_valid():
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(12)
    audio = rng.normal(0, 1, [2, 3000]).astype(np.float32)
    ir = (rng.normal(0, 1, [2000]) * np.exp(-np.arange(2000) / 300.0)).astype(np.float32)
    ref = O.fdn_get_signal(audio, ir)
    got = dp.FeedbackDelayNetworkApply().get_signal(_dev(audio), _dev(ir)).cpu().numpy()
    assert rms_err(got, ref) < TOL * rms(ref)
    refv = O.fft_convolve(audio, np.tile(ir[None], [2, 1]), padding='valid', delay_compensation=0)
    gotv = core.fft_convolve(_dev(audio), _dev(np.tile(ir[None], [2, 1])), padding='valid',
                             delay_compensation=0).cpu().numpy()
    assert gotv.shape == refv.shape == (2, 4999)
    assert rms_err(gotv, refv) < TOL * rms(refv)
    with pytest.raises(ValueError):
        core.fft_convolve(_dev(audio), _dev(np.zeros([3, 10], np.float32)))
    with pytest.raises(ValueError):
        core.fft_convolve(_dev(audio), _dev(np.zeros([2, 1999, 10], np.float32)))   # frame-count mismatch


@pytest.mark.parametrize('K,scale', [(96, 'exp_sigmoid'), (64, 'exp_tanh'), (32, 'none')])
def test_scale_fn_fused_into_the_fir_design_is_bit_identical(K, scale):
    """Audio-only route: FilteredNoise.get_controls' scale_fn(magnitudes + bias) runs inside the FIR design
    kernel instead of as its own pass over the [R, T, K] tensor."""
    import functools
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(3)
    raw = torch.as_tensor(rng.normal(0, 3, [5, 37, K]).astype(np.float32), device='cuda')
    fn = {'exp_sigmoid': dp.exp_sigmoid, 'exp_tanh': functools.partial(dp.exp_tanh, gain=0.7), 'none': None}[scale]
    synth = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=24000, scale_fn=fn, initial_bias=-2.5)
    rs = synth.raw_scale()
    if fn is None:
        assert rs is None
        return
    scaled = synth.get_controls(raw)['magnitudes']
    want = core.frequency_impulse_response(scaled, window_size=synth.window_size)
    got = core.frequency_impulse_response(raw, window_size=synth.window_size, raw_scale=rs)
    assert torch.equal(got, want)
    noise = torch.as_tensor(rng.uniform(-1, 1, [5, 37 * 96]).astype(np.float32), device='cuda')
    assert torch.equal(core.frequency_filter(noise, raw, window_size=synth.window_size, raw_scale=rs),
                       core.frequency_filter(noise, scaled, window_size=synth.window_size))
    # python callables that the library does not know stay outside the kernel
    assert dp.DynamicSizeFilteredNoise(scale_fn=lambda x: torch.sigmoid(x)).raw_scale() is None


@pytest.mark.parametrize('R,T,P,vq,split', [(3, 40, 1, 1, False), (16, 61, 8, 8, False), (16, 61, 8, 4, True), (32, 30, 16, 8, True),
                                            (4, 750, 2, 2, False)])
def test_matrix_pipe_walk_equals_the_vector_walk(R, T, P, vq, split, monkeypatch):
    """DDSPP_WIN_MFMA=1 (noise_win.hip: the walk as sixteen 4 x 4 outer products per v_mfma_f32_4x4x1, a skewed schedule
    resolved at compile time for the 24 kHz shape) against the default vector walk: the same (tap, sample) products added
    up in another order -- equal to a few ulp of the running sums, rows summed over voices and the last voice apart
    included; and against the oracle's frequency_filter."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(R * 7 + T)
    U, K = 96, 96
    N = T * U
    raw = torch.as_tensor(rng.normal(0, 2, [R, T, K]).astype(np.float32), device='cuda')
    noise = torch.as_tensor(rng.uniform(-1, 1, [R, N]).astype(np.float32), device='cuda')
    synth = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=250 * U, initial_bias=-3.0)
    rs = synth.raw_scale()
    outs = {}
    for mw in (0, 1, 2):
        set_option(monkeypatch, 'DDSPP_WIN_MFMA', mw)
        if P == 1:
            outs[mw] = (core.frequency_filter(noise, raw, window_size=synth.window_size, raw_scale=rs),)
        else:
            res = core.frequency_filter_voice_sums(noise, raw, synth.window_size, rs, P, vq, False, split_last=split)
            outs[mw] = tuple(t for t in (res if isinstance(res, (tuple, list)) else (res,)) if torch.is_tensor(t))
    set_option(monkeypatch, 'DDSPP_WIN_MFMA')
    assert len(outs[0]) == len(outs[1]) == len(outs[2])
    for a, b, c in zip(outs[0], outs[1], outs[2]):
        assert a.shape == b.shape == c.shape
        assert not torch.equal(a, b) or a.abs().max().item() == 0.0          # (the switch did switch)
        assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item())
        assert torch.equal(b, c)                    # 2: the same walk with the next unit's design riding in it
    if P == 1:
        mags = synth.get_controls(raw)['magnitudes'].cpu().numpy()
        ref = O.frequency_filter(noise.cpu().numpy(), mags, window_size=synth.window_size)
        assert rms_err(outs[1][0].cpu().numpy(), ref) < 1e-5


@pytest.mark.parametrize('B,T,U,K', [(3, 40, 96, 96), (2, 25, 96, 64), (1, 300, 96, 96), (2, 30, 128, 32), (5, 11, 192, 96)])
def test_fused_frequency_filter_equals_the_two_kernel_form(B, T, U, K, monkeypatch):
    """ddspp_frequency_filter_eo (design on the matrix cores + time-varying FIR, impulse responses kept in LDS)
    against ddspp_fir_from_magnitudes_eo + ddspp_time_varying_fir: same bits, with and without the fused scale_fn."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core, _lib
    rng = np.random.default_rng(17)
    N = T * U
    raw = torch.as_tensor(rng.normal(0, 2, [B, T, K]).astype(np.float32), device='cuda')
    noise = torch.as_tensor(rng.uniform(-1, 1, [B, N]).astype(np.float32), device='cuda')
    synth = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=250 * U, initial_bias=-3.0)
    Lw = 2 * (K - 1)
    assert _lib.load().ddspp_frequency_filter_eo_supported(N, T, K, Lw, -1) == 1
    for rs in (None, synth.raw_scale()):
        mags = raw if rs is not None else synth.get_controls(raw)['magnitudes']
        fused = core.frequency_filter(noise, mags, window_size=synth.window_size, raw_scale=rs)
        set_option(monkeypatch, 'DDSPP_FIR_NO_FUSED', '1')
        assert _lib.load().ddspp_frequency_filter_eo_supported(N, T, K, Lw, -1) == 0
        split = core.frequency_filter(noise, mags, window_size=synth.window_size, raw_scale=rs)
        set_option(monkeypatch, 'DDSPP_FIR_NO_FUSED')
        assert fused.shape == split.shape == (B, N)
        assert torch.equal(fused, split), (B, T, U, K, rs is not None)


@pytest.mark.parametrize('R,T,U,K,P,vq,split,vm', [(3, 40, 96, 96, 1, 1, False, False), (16, 61, 96, 96, 8, 8, False, False),
                                                   (16, 61, 96, 96, 8, 4, True, True), (32, 30, 96, 96, 16, 8, True, False),
                                                   (4, 750, 96, 96, 2, 2, False, False), (6, 33, 64, 64, 3, 1, False, False),
                                                   (8, 20, 128, 32, 4, 2, True, True), (4, 25, 192, 96, 2, 2, False, False),
                                                   (4, 31, 128, 128, 4, 4, False, False), (5, 50, 32, 32, 1, 1, False, False)])
def test_noise_drawn_inside_the_filter_kernel(R, T, U, K, P, vq, split, vm, monkeypatch):
    """Round 6: without a `noise=` argument the windowed FilteredNoise kernel draws its U(-1, 1) numbers while staging them
    (ddspp_frequency_filter_eo_voices_drawn) -- the Philox counters of their place in a [R, N] tensor, so the result is the
    filter of ddspp_uniform_noise's tensor bit for bit, which never has to exist.  Every hop / band count the kernel has an
    instance for, plain rows and voice sums, the last voice apart, both row orders, ragged windows (T no multiple of 30)."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core, _lib
    rng = np.random.default_rng(R * 31 + T)
    N = T * U
    raw = torch.as_tensor(rng.normal(0, 2, [R, T, K]).astype(np.float32), device='cuda')
    synth = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=250 * U, initial_bias=-3.0)
    rs = synth.raw_scale()
    assert _lib.load().ddspp_frequency_filter_eo_drawn_supported(N, T, K, 2 * (K - 1), -1) == 1
    seed, off = 0x1234567, (5 << 40) + 12
    lazy = core.DrawnNoise(R, N, seed, off, raw.device)
    tensor = core.uniform_noise((R, N), seed=seed, offset=off, device=raw.device)
    assert torch.equal(lazy.materialise(), tensor)
    if P == 1:
        a = (core.frequency_filter(lazy, raw, window_size=synth.window_size, raw_scale=rs),)
        b = (core.frequency_filter(tensor, raw, window_size=synth.window_size, raw_scale=rs),)
    else:
        a = core.frequency_filter_voice_sums(lazy, raw, synth.window_size, rs, P, vq, vm, split_last=split)
        b = core.frequency_filter_voice_sums(tensor, raw, synth.window_size, rs, P, vq, vm, split_last=split)
        a, b = (a if split else (a,)), (b if split else (b,))
    for x, y in zip(a, b):
        assert x.shape == y.shape and torch.equal(x, y) and float(y.abs().max()) > 0
    # the switch back (A/B): the tensor is drawn and read
    set_option(monkeypatch, 'DDSPP_NOISE_NO_DRAW', 1)
    assert _lib.load().ddspp_frequency_filter_eo_drawn_supported(N, T, K, 2 * (K - 1), -1) == 0
    c = core.frequency_filter(lazy, raw, window_size=synth.window_size, raw_scale=rs)
    set_option(monkeypatch, 'DDSPP_NOISE_NO_DRAW')
    assert torch.equal(c, core.frequency_filter(tensor, raw, window_size=synth.window_size, raw_scale=rs))


def test_processor_draws_the_same_numbers_either_way(monkeypatch):
    """DynamicSizeFilteredNoise()(magnitudes) and the batched group without `noise=`: the call counter advances as before and the
    audio equals the one made from the tensor the same (seed, call) would have drawn."""
    import ddsp_piano_amd as dp
    from ddsp_piano_amd import core
    rng = np.random.default_rng(77)
    B, T, K, sr = 3, 45, 96, 24000
    raw = torch.as_tensor(rng.normal(0, 1, [B, T, K]).astype(np.float32), device='cuda')
    g1 = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=sr)
    g2 = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=sr)
    g2.seed = g1.seed
    for _ in range(2):
        got = g1(raw)
        noise = g2.draw_noise(B, T * 96, raw.device)
        want = g2.get_signal(g2.get_controls(raw)['magnitudes'], noise=noise)
        assert torch.equal(got, want)
    # a shape the windowed kernel has no instance for (hop 80): the tensor is drawn, same numbers
    g3 = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=20000)
    g4 = dp.DynamicSizeFilteredNoise(frame_rate=250, sample_rate=20000)
    g4.seed = g3.seed
    got = g3(raw)
    want = g4.get_signal(g4.get_controls(raw)['magnitudes'], noise=g4.draw_noise(B, T * 80, raw.device))
    assert torch.equal(got, want)


def test_philox_known_answers():
    """The library's generator IS Philox4x32-10 (Salmon et al., SC'11): counter (0, 0, 0, 0) under key (0, 0) gives
    6627e8d5 e169c58d bc57ac4c 9b00dbd8 (Random123's known-answer vector), reachable through ddspp_uniform_noise's
    (offset, seed) = (counter words 0-1, key); a number is (word >> 8) 2^-23 - 1.  A plain-integer restatement of the ten
    rounds, checked against that vector, then gives the expected words for a second (offset, seed).  Holds whichever
    instruction sequence forms the 64-bit products (round 6: v_mad_u64_u32 instead of v_mul_hi_u32 + v_mul_lo_u32)."""
    from ddsp_piano_amd import core

    def want(words):
        return np.array([(w >> 8) * np.float32(2.0 / 16777216.0) - np.float32(1.0) for w in words], dtype=np.float32)

    got = core.uniform_noise((4,), seed=0, offset=0).cpu().numpy()
    assert np.array_equal(got, want([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]))
    # (counter words 2, 3 are always zero here, so Random123's other vectors are out of reach)
    def philox(ctr, key):
        c = [ctr & 0xffffffff, ctr >> 32, 0, 0]
        k = [key & 0xffffffff, key >> 32]
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
            k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
        return c
    assert philox(0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    seed, off = 0xa4093822299f31d0, 0x85a308d3243f6a88
    got = core.uniform_noise((8,), seed=seed, offset=off).cpu().numpy()
    assert np.array_equal(got, want(philox(off, seed) + philox(off + 1, seed)))
