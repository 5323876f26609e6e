"""CPU: the C-ABI library loads, exports every symbol include/ddspp.h declares, and rejects bad
arguments with an error code + message (no kernel is launched, no GPU needed)."""
import ctypes
import os
import re

from ddsp_piano_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'ddspp.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ddspp_[a-z0-9_]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/ddspp.h but not exported by libddspp.so'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature'
    assert sorted(_lib.SIGNATURES) == names


def test_identity(lib):
    assert lib.ddspp_version() >= 100
    assert lib.ddspp_target_arch() == b'gfx950'


def test_fft_size(lib):
    assert lib.ddspp_fft_size(72000, 72000) == 262144       # BASELINE config 2: 3 s audio, 3 s IR
    assert lib.ddspp_fft_size(72000, 48000) == 131072       # maestro-v2: L = 2 * sr
    assert lib.ddspp_fft_size(144000, 480000) == 1048576    # config 5
    assert lib.ddspp_fft_size(96, 190) == 512               # FilteredNoise frame (reference FFT size)
    assert lib.ddspp_fft_size(1, 1) == 1


def test_argument_errors_do_not_reach_the_gpu(lib):
    null = ctypes.c_void_p(0)
    rc = lib.ddspp_resample_linear(null, null, null, null, null, 1, 1, 1, 1, null)
    assert rc == _lib.DDSPP_EINVAL and b'null' in lib.ddspp_last_error()
    one = ctypes.c_void_p(16)
    rc = lib.ddspp_cos_oscillator_bank(one, one, one, 1, 12, 4, 24000.0, 1, 1, 0, null, 0, null)
    assert rc == _lib.DDSPP_EINVAL and b'multiple of 8' in lib.ddspp_last_error()
    rc = lib.ddspp_cos_oscillator_bank(one, one, one, 1, 16, 1000, 24000.0, 1, 1, 0, null, 0, null)
    assert rc == _lib.DDSPP_EINVAL and b'exceeds' in lib.ddspp_last_error()
    rc = lib.ddspp_harmonic_synthesis(one, one, one, one, one, one, one, 1, 10, 1, 8, 100, 24000.0, 1, 0, null, 0, null)
    assert rc == _lib.DDSPP_EINVAL and b'upsampling' in lib.ddspp_last_error()
    rc = lib.ddspp_time_varying_fir(one, one, one, 1, 100, 7, 10, -1, null)
    assert rc == _lib.DDSPP_EINVAL
    handle = ctypes.c_void_p()
    rc = lib.ddspp_fftconv_plan_create(4, 3, 100, 10, ctypes.byref(handle))
    assert rc == _lib.DDSPP_EINVAL and b'must be the same' in lib.ddspp_last_error()
    assert lib.ddspp_osc_workspace_bytes(0, 10, 10) == 0
    assert lib.ddspp_osc_workspace_bytes(1024, 72000, 128) > 0
    try:
        _lib.check(_lib.DDSPP_EINVAL)
    except ValueError:
        pass
    else:
        raise AssertionError('EINVAL must map to ValueError')


def test_argument_errors_of_the_surrogate_entry_points(lib):
    """SurrogateAdditive's entry points (late round 4) refuse bad arguments before any launch."""
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(256)
    rc = lib.ddspp_surrogate_harmonic_synthesis(one, one, one, one, null, null, one, one, one, 1, 10, 8, 96, 24000.0, 1, 0,
                                                null, 0, null)
    assert rc == _lib.DDSPP_EINVAL and b'decay' in lib.ddspp_last_error()
    rc = lib.ddspp_surrogate_harmonic_synthesis(one, one, one, one, one, one, one, one, one, 1, 10, 8, 100, 24000.0, 1, 0,
                                                null, 0, null)
    assert rc == _lib.DDSPP_EINVAL and b'upsampling' in lib.ddspp_last_error()
    rc = lib.ddspp_surrogate_decays(null, one, one, one, 1, 10, 8, 24000.0, null)
    assert rc == _lib.DDSPP_EINVAL and b'null' in lib.ddspp_last_error()
    rc = lib.ddspp_surrogate_decays(one, one, one, one, 1, 0, 8, 24000.0, null)
    assert rc == _lib.DDSPP_EINVAL and b'bad dims' in lib.ddspp_last_error()
    rc = lib.ddspp_polyphonic_surrogate_additive(one, one, one, null, one, null, null, null, one, one, one, null, 1, 2, 10, 8,
                                                 96, 24000.0, 0, 0, one, 1 << 20, null)
    assert rc == _lib.DDSPP_EINVAL and b'decay' in lib.ddspp_last_error()
    # normalisation mode of ddspp_inharmonic_controls: 0 before the cut, 1 after it, 2 never -- nothing else
    rc = lib.ddspp_inharmonic_controls(one, one, one, one, one, one, one, None, 1, 10, 8, 1, 24000.0, 20.0, 1, 10.0, 2.0, 1e-7,
                                       1.0, 3, 1, null)
    assert rc == _lib.DDSPP_EINVAL and b'normalize_after_nyquist_cut' in lib.ddspp_last_error()


def test_argument_errors_of_the_fused_and_split_entry_points(lib):
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(256)
    # shapes the fused FilteredNoise kernel takes / refuses (host-side geometry, no launch)
    assert lib.ddspp_frequency_filter_eo_supported(72000, 750, 96, 190, -1) == 1       # 24 kHz, maestro-v2
    assert lib.ddspp_frequency_filter_eo_supported(48000, 750, 64, 126, -1) == 1       # 16 kHz, dafx22 (round 3: the windowed kernel)
    assert lib.ddspp_frequency_filter_eo_supported(24000, 750, 32, 62, -1) == 1        # 8 kHz, ENSTDkCl-8kHz (hop 32)
    assert lib.ddspp_frequency_filter_eo_supported(96000, 750, 128, 254, -1) == 1      # 32 kHz, ENSTDkCl-32kHz (two workgroups per CU)
    assert lib.ddspp_frequency_filter_eo_supported(30000, 750, 32, 62, -1) == 0        # hop 40: no instance, two-call form
    assert lib.ddspp_frequency_filter_eo_supported(72000, 750, 96, 100, -1) == 0       # cropped window
    assert lib.ddspp_frequency_filter_eo_supported(72001, 750, 96, 190, -1) == 0
    rc = lib.ddspp_frequency_filter_eo(null, one, one, one, one, one, one, one, 4, 72000, 750, 96, 190, 48, -1, 1,
                                       -5.0, 10.0, 2.0, 1e-7, 1.0, null)
    assert rc == _lib.DDSPP_EINVAL and b'null' in lib.ddspp_last_error()
    rc = lib.ddspp_frequency_filter_eo_voices(one, one, one, one, one, one, one, one, null, 32, 72000, 750, 96, 190, 48, -1,
                                              1, -5.0, 10.0, 2.0, 1e-7, 1.0, 16, 3, 0, null)      # 3 does not divide 16
    assert rc == _lib.DDSPP_EINVAL and b'voices' in lib.ddspp_last_error()
    rc = lib.ddspp_frequency_filter_eo(one, one, one, one, one, one, one, one, 4, 30000, 750, 32, 62, 16, -1, 1,
                                       -5.0, 10.0, 2.0, 1e-7, 1.0, null)
    assert rc == _lib.DDSPP_EINVAL and b'not supported' in lib.ddspp_last_error()
    assert lib.ddspp_fftconv_transform_ir(null, one, 1, one, 0, null) == _lib.DDSPP_EINVAL
    assert lib.ddspp_fftconv_execute_prepared(null, one, 10, one, 10, 0, 1, one, 0, null) == _lib.DDSPP_EINVAL
    rc = lib.ddspp_polyphonic_additive(one, one, one, one, null, null, one, one, null, one, null, 2, 65, 10, 1, 8, 96, 24000.0,
                                       0, 0, one, 1 << 30, null)
    assert rc == _lib.DDSPP_EINVAL and b'exceeds 64' in lib.ddspp_last_error()


def test_host_table_builders_agree_with_the_python_layer():
    """csrc/tables.cpp (what a caller without the Python layer uses) against core.py's numpy tables: integer tables and
    the pure float32 products are identical; cosine-derived entries may differ by an ulp of a libm."""
    import ctypes

    import numpy as np

    from ddsp_piano_amd import _lib, core
    lib = _lib.load()

    def ptr(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    for n in (2, 64, 192, 257, 384, 190):
        w = np.empty(n, np.float32)
        assert lib.ddspp_hann_window_host(n, ptr(w)) == 0
        assert np.abs(w - core._hann_window_np(n)).max() <= 1.2e-7
    for rule, name in ((0, 'legacy'), (1, 'half_pixel')):
        for T, N in ((750, 72000), (37, 1000), (12, 12 * 64), (34000, 34000 * 96)):
            lo, hi, w = np.empty(N, np.int32), np.empty(N, np.int32), np.empty(N, np.float32)
            al = ctypes.c_int(-1)
            assert lib.ddspp_resample_tables_host(T, N, rule, ptr(lo), ptr(hi), ptr(w), ctypes.byref(al)) == 0
            plo, phi, pw, pal = core._linear_tables_np(T, N, name)
            assert np.array_equal(lo, plo) and np.array_equal(hi, phi) and np.array_equal(w, pw) and bool(al.value) == pal
            # a streamed piece: the weights of absolute positions = the matching slice of the long signal's table,
            # which is what the host layer builds with torch (core.linear_weights)
            if N % T == 0 and N >= 64:
                first, n = N // 2, N // 4
                wp = np.empty(n, np.float32)
                assert lib.ddspp_linear_weights_host(T, N, rule, first, n, ptr(wp)) == 0
                assert np.array_equal(wp, pw[first:first + n])
                prev = core.set_recalled(resize=name)
                try:
                    tp = max(T // 4, 1)
                    wt = core.linear_weights(tp, tp * (N // T), 'cpu', first).numpy()
                finally:
                    core.set_recalled(**prev)
                m = min(n, wt.size)
                assert np.array_equal(wt[:m], pw[first:first + m])
    for K, ws in ((96, 257), (64, 257), (32, 257), (128, 257), (200, 257), (65, 0), (129, 257), (200, 101)):
        lw, nj = ctypes.c_int(), ctypes.c_int()
        assert lib.ddspp_fir_tables_shape(K, ws, ctypes.byref(lw), ctypes.byref(nj)) == 0
        for crop, cname in ((0, 'ddsp370'), (1, 'centred')):
            pm = core._fir_matrix_np(K, ws, cname)
            assert pm.shape == (K, lw.value)
            m = np.empty((K, lw.value), np.float32)
            uq, mr, nu = np.empty(lw.value, np.int32), np.empty(lw.value, np.int32), ctypes.c_int()
            assert lib.ddspp_fir_matrix_host(K, ws, crop, ptr(m), ptr(uq), ptr(mr), ctypes.byref(nu)) == 0
            assert np.abs(m - pm).max() <= 2e-7 * np.abs(pm).max()
            puq, pmr = core._fir_symmetry_np(K, ws, cname)
            assert np.array_equal(uq[:nu.value], puq) and np.array_equal(mr[:nu.value], pmr)
        eo = core._fir_eo_tables_np(K, ws)
        assert (eo is None) == (nj.value == 0)
        if eo is not None:
            ce, co, idx, we, wo, pnj, plw = eo
            assert pnj == nj.value and plw == lw.value
            c_ce, c_co = np.empty_like(ce), np.empty_like(co)
            c_idx, c_we, c_wo = np.empty_like(idx), np.empty_like(we), np.empty_like(wo)
            assert lib.ddspp_fir_eo_tables_host(K, ws, ptr(c_ce), ptr(c_co), ptr(c_idx), ptr(c_we), ptr(c_wo)) == 0
            assert np.array_equal(c_idx, idx)
            for a, b in ((c_ce, ce), (c_co, co), (c_we, we), (c_wo, wo)):
                assert np.abs(a - b).max() <= 1.2e-7 * max(1.0, float(np.abs(b).max()))
    assert lib.ddspp_fir_eo_tables_host(65, 0, None, None, None, None, None) == _lib.DDSPP_EINVAL


def test_options_are_read_once_and_reloadable(monkeypatch):
    """No getenv on the call path: an option's environment variable is consulted on first use and cached;
    ddspp_set_option overrides it, ddspp_reload_options forgets the cache.  The host layer's switches likewise."""
    from ddsp_piano_amd import _lib
    lib = _lib.load()
    monkeypatch.setenv('DDSPP_TEST_OPTION', '5')
    lib.ddspp_reload_options()
    assert lib.ddspp_option(b'DDSPP_TEST_OPTION', 1) == 5
    monkeypatch.setenv('DDSPP_TEST_OPTION', '7')
    assert lib.ddspp_option(b'DDSPP_TEST_OPTION', 1) == 5            # cached
    assert lib.ddspp_set_option(b'DDSPP_TEST_OPTION', 9) == 0
    assert lib.ddspp_option(b'DDSPP_TEST_OPTION', 1) == 9
    lib.ddspp_reload_options()
    assert lib.ddspp_option(b'DDSPP_TEST_OPTION', 1) == 7
    monkeypatch.delenv('DDSPP_TEST_OPTION')
    lib.ddspp_reload_options()
    assert lib.ddspp_option(b'DDSPP_TEST_OPTION', 1) == 1
    assert lib.ddspp_set_option(None, 1) == _lib.DDSPP_EINVAL
    _lib.options.reload()
    monkeypatch.setenv('DDSPP_SIDE_STREAM', '1')                     # the second stream is opt-in
    assert _lib.options.side_stream is False                         # not re-read per call
    _lib.options.reload()
    assert _lib.options.side_stream is True and _lib.options.no_side_stream is False
    monkeypatch.delenv('DDSPP_SIDE_STREAM')
    _lib.options.reload()
    assert _lib.options.no_side_stream is True


def test_group_driver_argument_errors(lib):
    """ddspp_group_create validates the configuration before it touches the device; ddspp_group_run refuses null
    arguments; the struct the Python layer passes has the layout of the header's (size check through a known field)."""
    import ctypes
    from ddsp_piano_amd import _lib
    from ddsp_piano_amd.native_group import _Config, _Outputs
    h = ctypes.c_void_p()
    assert lib.ddspp_group_create(None, ctypes.byref(h)) == _lib.DDSPP_EINVAL
    c = _Config()
    assert lib.ddspp_group_create(ctypes.byref(c), ctypes.byref(h)) == _lib.DDSPP_EINVAL          # all-zero dimensions
    assert b'bad dimensions' in lib.ddspp_last_error()
    c.n_segments, c.n_voices, c.n_frames, c.n_harmonics, c.n_substrings, c.n_bands, c.upsampling = 2, 16, 50, 128, 1, 96, 100
    assert lib.ddspp_group_create(ctypes.byref(c), ctypes.byref(h)) == _lib.DDSPP_EINVAL
    assert b'multiple of 8' in lib.ddspp_last_error()
    c.upsampling, c.n_voices, c.n_substrings = 96, 40, 2
    assert lib.ddspp_group_create(ctypes.byref(c), ctypes.byref(h)) == _lib.DDSPP_EINVAL
    assert b'exceeds 64' in lib.ddspp_last_error()
    c.n_voices, c.n_substrings, c.ir_batch = 16, 1, 3
    assert lib.ddspp_group_create(ctypes.byref(c), ctypes.byref(h)) == _lib.DDSPP_EINVAL
    assert b'ir_batch' in lib.ddspp_last_error()
    c.ir_batch, c.n_bands, c.window_size = 0, 65, 257
    assert lib.ddspp_group_create(ctypes.byref(c), ctypes.byref(h)) == _lib.DDSPP_EINVAL            # no even/odd tables
    assert b'even/odd' in lib.ddspp_last_error()
    assert lib.ddspp_group_run(None, None, None, None, None, None, None, None, None, None, None, 0, None) == _lib.DDSPP_EINVAL
    assert lib.ddspp_group_workspace_bytes(None) == 0 and lib.ddspp_group_n_samples(None) == -1
    lib.ddspp_group_destroy(None)                                                                   # a no-op
    # 11 ints, 2 floats, 1 int, 4 floats, 4 ints, 5 floats, 2 ints, (pad), uint64, 2 ints -- as include/ddspp.h declares them
    assert ctypes.sizeof(_Config) == 29 * 4 + 4 + 8 + 8 and _Config.noise_seed.offset == 120
    assert _Config.reverb_keep_dry_tap.offset == 128
    assert ctypes.sizeof(_Config) == lib.ddspp_group_config_bytes()
    assert ctypes.sizeof(_Outputs) == lib.ddspp_group_outputs_bytes() == 8 * ctypes.sizeof(ctypes.c_void_p)


def test_round6_entry_points_refuse_what_they_cannot_do(lib):
    """ddspp_inharmonic_controls_sparse needs the counts it tells its reader about; the drawn-noise filter exists only for the
    shapes of the windowed kernel (host-side checks, no launch)."""
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(256)
    rc = lib.ddspp_inharmonic_controls_sparse(one, one, one, one, one, one, null, null, 32, 750, 128, 1, 16, 0, 24000.0, 20.0, 1,
                                              10.0, 2.0, 1e-7, 1.0, 1, 1, null)
    assert rc == _lib.DDSPP_EINVAL and b'audible_out' in lib.ddspp_last_error()
    assert lib.ddspp_frequency_filter_eo_drawn_supported(72000, 750, 96, 190, -1) == 1       # 24 kHz
    assert lib.ddspp_frequency_filter_eo_drawn_supported(144000, 750, 96, 190, -1) == 1      # 48 kHz, hop 192
    assert lib.ddspp_frequency_filter_eo_drawn_supported(30000, 750, 32, 62, -1) == 0        # hop 40: no windowed instance
    rc = lib.ddspp_frequency_filter_eo_voices_drawn(1, 0, one, one, one, one, one, one, one, null, 4, 30000, 750, 32, 62, 16, -1, 1,
                                                    -5.0, 10.0, 2.0, 1e-7, 1.0, 1, 1, 0, null)
    assert rc == _lib.DDSPP_EINVAL and b'windowed kernel' in lib.ddspp_last_error()
    rc = lib.ddspp_frequency_filter_eo_voices_drawn(1, 0, null, one, one, one, one, one, one, null, 4, 72000, 750, 96, 190, 48, -1, 1,
                                                    -5.0, 10.0, 2.0, 1e-7, 1.0, 1, 1, 0, null)
    assert rc == _lib.DDSPP_EINVAL and b'null' in lib.ddspp_last_error()
