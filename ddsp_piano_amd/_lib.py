"""Build + ctypes binding of ``libddspp.so`` (the hand-written HIP kernels, C-ABI in include/ddspp.h).

PyTorch-ROCm is only the buffer carrier: every call passes raw device pointers and the current HIP
stream.  There is no CPU fallback -- if the shared library is missing or cannot be loaded the
import of any operator fails loudly.
"""
from __future__ import annotations

import concurrent.futures
import ctypes
import os
import shutil
import subprocess
import sys
from ctypes import c_char_p, c_float, c_int, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(_HERE, 'libddspp.so')
SOURCES = ['error.cpp', 'midi_conditioning.cpp', 'tables.cpp', 'oscillator.hip', 'oscillator_p1.hip', 'oscillator_p2.hip', 'oscillator_p3.hip', 'osc_stream.hip', 'bank_compact.hip', 'resample.hip', 'controls.hip',
           'noise.hip', 'noise_win.hip', 'noise_bands.hip', 'reverb.hip', 'reverb_part.hip', 'fdn.hip', 'probe.hip', 'group.cpp']
ARCH = 'gfx950'
HEADERS = ['ddspp_common.h', 'osc_common.h', 'noise_win.h', 'reverb_part.h', os.path.join('..', '..', 'include', 'ddspp.h')]
# packed f32 math has the per-element rate of plain VALU ops on gfx950 (profiles/r01_ubench.txt); in the
# time-varying FIR the SLP vectoriser's v_pk_fma_f32 operand pairs cost a dozen extra LDS reads / moves per step
PER_FILE_FLAGS = {'noise.hip': ['-fno-slp-vectorize'], 'noise_win.hip': ['-fno-slp-vectorize'], 'oscillator.hip': ['-fno-slp-vectorize'], 'oscillator_p1.hip': ['-fno-slp-vectorize'], 'oscillator_p2.hip': ['-fno-slp-vectorize'],
                  'oscillator_p3.hip': ['-fno-slp-vectorize'], 'osc_stream.hip': ['-fno-slp-vectorize'],
                  'bank_compact.hip': ['-fno-slp-vectorize']}

# sources a translation unit includes beside the HEADERS (oscillator_p*.hip are oscillator.hip again, with DDSPP_OSC_PART set)
EXTRA_DEPS = {f'oscillator_p{k}.hip': ['oscillator.hip'] for k in (1, 2, 3)}

DDSPP_OK = 0
DDSPP_EINVAL = -22


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: cannot build libddspp.so')


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources and headers: profiles/*.json that hold counter values of
    the kernels carry it, so that bench.py can tell when they describe another build (counters_stale)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES) + sorted(HEADERS):
        if name == 'probe.hip':            # measurement aids (read / multiply-add probes): no kernel of the path lives there
            continue
        with open(os.path.join(_CSRC, name), 'rb') as f:
            h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def _needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(_CSRC, s) for s in SOURCES] + [os.path.join(_CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP translation unit for gfx950 and link libddspp.so in-tree."""
    if not force and not _needs_build():
        return LIB_PATH
    hipcc = _hipcc()
    objdir = os.path.join(_HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    flags = [f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC',
             '-I', _CSRC, '-Wno-unused-value']

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + '.o')
        srcp = os.path.join(_CSRC, src)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(srcp)
                and all(os.path.getmtime(obj) > os.path.getmtime(os.path.join(_CSRC, h)) for h in HEADERS + EXTRA_DEPS.get(src, []))):
            return obj
        extra = PER_FILE_FLAGS.get(src, [])
        cmd = [hipcc] + flags + extra + ['-x', 'hip', '-c', srcp, '-o', obj]
        if verbose:
            print('[ddspp build]', ' '.join(cmd), file=sys.stderr, flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), 'lib')
    tmp = LIB_PATH + '.tmp'
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC'] + objs + \
          ['-L', rocm_lib, '-L', '/opt/rocm/lib', '-lrocfft', '-Wl,-rpath,/opt/rocm/lib', '-o', tmp]
    if verbose:
        print('[ddspp build]', ' '.join(cmd), file=sys.stderr, flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


# name -> (restype, argtypes); must list every symbol declared in include/ddspp.h
SIGNATURES = {
    'ddspp_version': (c_int, []),
    'ddspp_target_arch': (c_char_p, []),
    'ddspp_last_error': (c_char_p, []),
    'ddspp_option': (c_int, [c_char_p, c_int]),
    'ddspp_set_option': (c_int, [c_char_p, c_int]),
    'ddspp_reload_options': (None, []),
    'ddspp_hann_window_host': (c_int, [c_int, c_void_p]),
    'ddspp_resample_tables_host': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ddspp_group_config_bytes': (ctypes.c_size_t, []),
    'ddspp_group_outputs_bytes': (ctypes.c_size_t, []),
    'ddspp_group_create': (c_int, [c_void_p, c_void_p]),
    'ddspp_group_destroy': (None, [c_void_p]),
    'ddspp_group_workspace_bytes': (ctypes.c_size_t, [c_void_p]),
    'ddspp_group_n_samples': (c_int, [c_void_p]),
    'ddspp_group_run': (c_int, [c_void_p] * 11 + [ctypes.c_size_t, c_void_p]),
    'ddspp_linear_weights_host': (c_int, [c_int, c_int, c_int, ctypes.c_longlong, c_int, c_void_p]),
    'ddspp_hbm_read_probe': (c_int, [c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    'ddspp_hbm_write_probe': (c_int, [c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]),
    'ddspp_fma_probe': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'ddspp_walk_weights_host': (c_int, [c_int, c_int, c_int, ctypes.c_longlong, c_int, c_void_p, c_void_p]),
    'ddspp_fir_tables_shape': (c_int, [c_int, c_int, c_void_p, c_void_p]),
    'ddspp_fir_matrix_host': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ddspp_fir_eo_tables_host': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'ddspp_resample_linear': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                      c_int, c_int, c_void_p]),
    'ddspp_resample_window': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ddspp_decay_envelope': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ddspp_osc_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'ddspp_cos_oscillator_bank': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                          c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'ddspp_harmonic_synthesis': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                         c_int, c_void_p, c_size_t, c_void_p]),
    'ddspp_surrogate_harmonic_synthesis': (c_int, [c_void_p] * 9 + [c_int] * 4 + [c_float, c_int, c_int, c_void_p, c_size_t,
                                                   c_void_p]),
    'ddspp_surrogate_decays': (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_float, c_void_p]),
    'ddspp_polyphonic_additive_workspace_bytes': (c_size_t, [c_int] * 6),
    'ddspp_polyphonic_additive': (c_int, [c_void_p] * 11 + [c_int] * 6 + [c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'ddspp_polyphonic_stems_workspace_bytes': (c_size_t, [c_int] * 6),
    'ddspp_polyphonic_stems': (c_int, [c_void_p] * 9 + [c_int] * 6 + [c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'ddspp_polyphonic_surrogate_additive': (c_int, [c_void_p] * 12 + [c_int] * 5 + [c_float, c_int, c_int, c_void_p, c_size_t,
                                                    c_void_p]),
    'ddspp_oscillator_phase_state_workspace_bytes': (c_size_t, [c_int] * 4),
    'ddspp_oscillator_phase_state': (c_int, [c_void_p] * 8 + [c_int] * 5 + [c_float, c_int, c_void_p, c_size_t, c_void_p]),
    'ddspp_inharmonic_controls': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_int,
                                          c_float, c_float, c_float, c_float, c_int, c_int, c_void_p]),
    'ddspp_inharmonic_controls_group': (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_float, c_float, c_int, c_float, c_float, c_float,
                                                c_float, c_int, c_int, c_void_p]),
    'ddspp_inharmonic_controls_sparse': (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_float, c_float, c_int, c_float, c_float, c_float,
                                                 c_float, c_int, c_int, c_void_p]),
    'ddspp_scale_bias': (c_int, [c_void_p, c_void_p, c_size_t, c_float, c_int, c_float, c_float, c_float,
                                 c_float, c_void_p]),
    'ddspp_add_signals': (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'ddspp_polyphonic_mix': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'ddspp_add_chain_paired': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ddspp_mix_voices': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ddspp_mix_last_voice': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_void_p]),
    'ddspp_mix_last_voice_paired': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_int, c_int, c_int, c_void_p]),
    'ddspp_fir_from_magnitudes': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                          c_int, c_int, c_void_p]),
    'ddspp_fir_from_magnitudes_eo': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_size_t, c_int, c_int, c_int, c_int, c_float, c_float, c_float,
                                             c_float, c_float, c_void_p]),
    'ddspp_frequency_filter_eo_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'ddspp_frequency_filter_eo': (c_int, [c_void_p] * 8 + [c_int] * 8 + [c_float] * 5 + [c_void_p]),
    'ddspp_frequency_filter_eo_voices': (c_int, [c_void_p] * 9 + [c_int] * 8 + [c_float] * 5 + [c_int] * 3 + [c_void_p]),
    'ddspp_frequency_filter_eo_drawn_supported': (c_int, [c_int] * 5),
    'ddspp_frequency_filter_eo_voices_drawn': (c_int, [c_uint64, c_uint64] + [c_void_p] * 8 + [c_int] * 8 + [c_float] * 5 +
                                               [c_int] * 3 + [c_void_p]),
    'ddspp_time_varying_fir': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_void_p]),
    'ddspp_noise_bands': (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    'ddspp_uniform_noise': (c_int, [c_void_p, c_size_t, c_uint64, c_uint64, c_void_p]),
    'ddspp_uniform_noise_rows': (c_int, [c_void_p, c_int, c_size_t, c_uint64, c_uint64, c_uint64, c_void_p]),
    'ddspp_fft_size': (c_int, [c_int, c_int]),
    'ddspp_fftconv_plan_create': (c_int, [c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    'ddspp_fftconv_plan_destroy': (c_int, [c_void_p]),
    'ddspp_fftconv_workspace_bytes': (c_size_t, [c_void_p]),
    'ddspp_fftconv_fft_size': (c_int, [c_void_p]),
    'ddspp_fftconv_execute': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                      c_int, c_void_p, c_size_t, c_void_p]),
    'ddspp_fftconv_transform_ir': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'ddspp_fftconv_execute_prepared': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                               c_size_t, c_void_p]),
    'ddspp_fdn_transfer': (c_int, [c_void_p] * 9 + [c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    'ddspp_fdn_add_early': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'ddspp_irfft_plan_create': (c_int, [c_int, c_int, ctypes.POINTER(c_void_p)]),
    'ddspp_irfft_plan_destroy': (c_int, [c_void_p]),
    'ddspp_irfft_workspace_bytes': (c_size_t, [c_void_p]),
    'ddspp_irfft_execute': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ddspp_midi_conditioning_create': (c_void_p, [c_int]),
    'ddspp_midi_conditioning_destroy': (None, [c_void_p]),
    'ddspp_midi_conditioning_reset': (c_int, [c_void_p]),
    'ddspp_midi_conditioning_get_state': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'ddspp_midi_conditioning_run_f64': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'ddspp_midi_conditioning_run_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
}

_lib = None


def _have_hipcc():
    try:
        _hipcc()
        return True
    except RuntimeError:
        return False


def load():
    """Load libddspp.so.  A missing library is built; a library OLDER than its sources is rebuilt when hipcc is here
    (an edited kernel never runs as a stale binary) and refused with an error otherwise."""
    global _lib
    if _lib is not None:
        return _lib
    custom = os.environ.get('DDSPP_LIB')
    if not custom and _needs_build():
        missing = not os.path.exists(LIB_PATH)
        if _have_hipcc():
            try:
                build(verbose=False)
            except Exception as e:  # noqa: BLE001
                raise RuntimeError(
                    f'libddspp.so at {LIB_PATH} is {"missing" if missing else "older than csrc/"} and could not be built '
                    f'({e}).  The DDSP-Piano MI355X synthesis path has no CPU fallback: run '
                    '`python -c "import __graft_entry__ as g; g.build()"` on a machine with hipcc.') from e
        elif missing:
            raise RuntimeError(f'libddspp.so is missing at {LIB_PATH} and hipcc is not available to build it.  '
                               'There is no CPU fallback for the synthesis path.')
        else:
            raise RuntimeError(f'{LIB_PATH} is older than its sources under csrc/ and hipcc is not available to rebuild '
                               'it: refusing to run a stale kernel library.')
    path = custom or LIB_PATH                             # DDSPP_LIB: another build of the same sources (A/B timing)
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:
        raise RuntimeError(f'cannot load {path}: {e}. No CPU fallback exists.') from e
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


class Options:
    """Tuning / A-B switches of the host layer, read from the environment ONCE (import time) -- nothing on the call path
    reads os.environ.  reload() re-reads them and tells the library to forget its cached options too (tests and the A/B
    tools change variables in-process)."""

    def __init__(self):
        self.reload(_library=False)

    def reload(self, _library=True):
        env = os.environ
        self.voice_sums = int(env.get('DDSPP_VOICE_SUMS', 0))            # voices summed per noise row (0: pick)
        self.no_voice_sums = env.get('DDSPP_NO_VOICE_SUMS') == '1'
        # noise branch on a side stream: opt-in since round 3 (DDSPP_SIDE_STREAM=1).  Both branches are bound by the same
        # VALU issue slots; same-box A/B at batch 64: 1.96 ms with one stream against 1.99 ms with two
        self.side_stream = env.get('DDSPP_SIDE_STREAM') == '1'
        self.side_stream_min = int(env.get('DDSPP_SIDE_STREAM_MIN', 1 << 24))
        self.no_side_stream = env.get('DDSPP_NO_SIDE_STREAM') == '1' or not self.side_stream
        self.no_early_ir = env.get('DDSPP_NO_EARLY_IR') == '1'
        self.stems_single = env.get('DDSPP_STEMS_SINGLE') == '1'            # every voice's stems: every voice a segment of its own (A/B)
        self.no_stems_compact = env.get('DDSPP_NO_STEMS_COMPACT') == '1'    # every voice's stems through the per-voice fused kernel (A/B)
        self.surrogate_materialised = env.get('DDSPP_SURROGATE_MATERIALISED') == '1'   # SurrogateAdditive: the three-operator route (A/B)
        if _library and _lib is not None:
            _lib.ddspp_reload_options()
            for name, value in _persistent_options.items():         # settings that are not tuning (core.set_recalled)
                _lib.ddspp_set_option(name.encode(), int(value))


_persistent_options = {}


def set_option(name, value, persistent=False):
    """ddspp_set_option; persistent=True: re-applied after every Options.reload() (which makes the library forget its
    cache): the recalled-detail switches (core.set_recalled) are settings, not A/B variables."""
    if persistent:
        _persistent_options[name] = int(value)
    check(load().ddspp_set_option(name.encode(), int(value)))


options = Options()


def last_error():
    return load().ddspp_last_error().decode('utf-8', 'replace')


def check(rc):
    """Map a C-ABI return code onto the exception type ddsp raises for the same mistake."""
    if rc == DDSPP_OK:
        return
    msg = last_error()
    if rc == DDSPP_EINVAL:
        raise ValueError(msg)
    raise RuntimeError(f'libddspp error {rc}: {msg}')
