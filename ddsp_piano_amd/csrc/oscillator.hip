// Inharmonic additive oscillator bank for gfx950 (CDNA4).
//
// Replaces, behind the C-ABI declared in include/ddspp.h:
//   * cos_oscillator_bank            ddsp_piano/modules/inharm_synth.py:49-84
//       (+ ddsp.core.remove_above_nyquist, ddsp.core.angular_cumsum, tf.cumsum, tf.cos,
//        tf.reduce_sum underneath it)
//   * harmonic_synthesis             ddsp_piano/modules/inharm_synth.py:87-127
//       (+ ddsp.core.get_harmonic_frequencies, ddsp.core.resample 'linear' and 'window')
//   * the substring loop of MultiInharmonic.get_signal   inharm_synth.py:272-293
//
// Work decomposition:
//   one wavefront = one (row, span) task, row = batch x voice, span = a run of consecutive
//   1000-sample chunks of ddsp.core.angular_cumsum.  Lane l owns the "virtual oscillators"
//   v = l + 64 j (j < VPL), v = substring * H + harmonic, and walks time SEQUENTIALLY, so the
//   float32 phase accumulation has exactly the order of the reference's CPU scan.  Rows of the
//   [.., N, H] envelopes are read as 256-byte coalesced wave loads (harmonic = fastest axis).
//   The harmonic sum is transposed through LDS: every lane deposits its per-sample partial into a
//   [32 samples][64 lanes] tile (row stride 65 words: conflict free both ways), then each lane
//   sums one column and 32 lanes store 128 contiguous bytes of audio.
//   When rows alone cannot fill the chip the time axis is cut into spans; a cheap pre-pass
//   (phase only) produces each chunk's end phase and a tiny sequential scan turns them into the
//   exact float32 chunk offsets each span starts from.  One span per row (no pre-pass, envelopes
//   read exactly once) is the HBM-roofline configuration.
#include <type_traits>

#include "ddspp_common.h"

namespace ddspp {

struct OscParams {
    // materialised source: cos_oscillator_bank(frequency_envelopes, amplitude_envelopes)
    const float* __restrict__ fe;      // [R, N, H]
    const float* __restrict__ ae;      // [R, N, H]
    // fused source: frame-rate controls of harmonic_synthesis / MultiInharmonic.get_signal
    const float* __restrict__ f0;      // [R, T, S]
    const float* __restrict__ amp;     // [R, T]
    const float* __restrict__ hd;      // [R, T, H]
    const float* __restrict__ shifts;  // [R, T, H] (may be null: no shifts)
    const float* __restrict__ wlin;    // [N]   legacy-bilinear interpolation weight per sample
    const float* __restrict__ whann;   // [2U]  tf.signal.hann_window(2U)
    float* __restrict__ out;           // [R, N] (sum) or [R, N, V]
    float* __restrict__ ework;         // [R, npre, VP]  chunk end phase mod 2pi
    const float* __restrict__ astart;  // [R, spans, VP] running offset sum at span start
    int R, N, T, U, H, S, V, VP;
    int spans, cps, nchunks, npre;
    float sr, rsr, nyq;
    int fastdiv;                       // sample rate is in the exhaustively checked list
};

enum { MODE_MAIN = 0, MODE_PREPASS = 1, MODE_PLAIN = 2 };

constexpr int BLK = 8;        // samples per unrolled block; divides U and the 1000-sample chunk
constexpr int TILE = 32;      // samples per LDS reduction tile
constexpr int TSTRIDE = 65;   // words per tile row

template <bool FAST>
__device__ __forceinline__ float omega_of(float fe, float sr, float rsr) {
    float om = fe * DDSPP_TWO_PI_F32;             // inharm_synth.py:69
    if (FAST) return div_const(om, sr, rsr);      // inharm_synth.py:70, exact (see ddspp_common.h)
    return om / sr;
}

template <int VPL, bool FUSED, int MODE, bool SUM>
__global__ void __launch_bounds__(256) osc_kernel(const OscParams p) {
    __shared__ float lds_tile[4][TILE * TSTRIDE];

    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int task = wave_uniform(blockIdx.x * 4 + wib);
    int row, c0, c1;
    if (MODE == MODE_PREPASS) {
        if (task >= p.R * p.npre) return;
        row = task / p.npre;
        c0 = task - row * p.npre;
        c1 = c0 + 1;
    } else {
        if (task >= p.R * p.spans) return;
        row = task / p.spans;
        const int span = task - row * p.spans;
        c0 = span * p.cps;
        c1 = min(c0 + p.cps, p.nchunks);
    }
    float* tile = lds_tile[wib];

    const int N = p.N, U = p.U, H = p.H, T = p.T, S = p.S, V = p.V;
    const int n_begin = c0 * DDSPP_CHUNK;
    const int n_end = min(c1 * DDSPP_CHUNK, N);

    // ---- per-lane oscillator identity ---------------------------------------------------------
    int vk[VPL], vs[VPL];
    bool valid[VPL];
    float kmul[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int v = lane + 64 * j;
        valid[j] = v < V;
        vs[j] = valid[j] ? v / H : 0;
        vk[j] = valid[j] ? v - vs[j] * H : 0;
        kmul[j] = (float)(vk[j] + 1);              // linspace(1, H, H)
    }

    // ---- running state -------------------------------------------------------------------------
    float ph[VPL], asum[VPL], off[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        ph[j] = 0.0f;
        asum[j] = 0.0f;
        off[j] = 0.0f;
    }
    if (MODE == MODE_MAIN && p.spans > 1) {
        const int span = c0 / p.cps;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            asum[j] = p.astart[((size_t)row * p.spans + span) * p.VP + lane + 64 * j];
            off[j] = mod_2pi(asum[j]);
        }
    }

    // ---- fused source: frame controls ----------------------------------------------------------
    // hf(t, v) = (f0[t, s] * k) * (1 + shift[t, k])     inharm_synth.py:106-108
    // ha(t, v) = amp[t] * hd[t, k]                      inharm_synth.py:112
    float x0[VPL], x1[VPL], a0[VPL], a1[VPL];
    int t = 0, r = 0;
    auto frame_ctl = [&](int tt, float* xf, float* xa) {
        const float amp_t = p.amp[(size_t)row * T + tt];
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            float f = 0.0f, a = 0.0f;
            if (valid[j]) {
                const float f0t = p.f0[((size_t)row * T + tt) * S + vs[j]];
                f = f0t * kmul[j];
                if (p.shifts) f = f * (1.0f + p.shifts[((size_t)row * T + tt) * H + vk[j]]);
                if (MODE != MODE_PREPASS) a = amp_t * p.hd[((size_t)row * T + tt) * H + vk[j]];
            }
            xf[j] = f;
            xa[j] = a;
        }
    };
    if (FUSED) {
        t = n_begin / U;
        r = n_begin - t * U;
        frame_ctl(t, x0, a0);
        frame_ctl(min(t + 1, T - 1), x1, a1);
    }

    int cpos = 0;                      // position inside the current 1000-sample chunk
    int chunk = c0;
    const float* fe_row = FUSED ? nullptr : p.fe + (size_t)row * N * H;
    const float* ae_row = FUSED ? nullptr : p.ae + (size_t)row * N * H;

    // materialised source: register double buffer, next block prefetched while this one computes
    float fbuf[2][BLK][VPL], abuf[2][BLK][VPL];
    auto load_block = [&](int n0, float (*fb)[VPL], float (*ab)[VPL]) {
#pragma unroll
        for (int i = 0; i < BLK; ++i) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                float f = 0.0f, a = 0.0f;
                if (valid[j] && n0 + i < n_end) {
                    const size_t idx = (size_t)(n0 + i) * H + lane + 64 * j;
                    f = fe_row[idx];
                    if (MODE != MODE_PREPASS) a = ae_row[idx];
                }
                fb[i][j] = f;
                ab[i][j] = a;
            }
        }
    };

    // One block of BLK samples.  FDIV / FMOD select the exact fast forms of the constant division
    // and of the 2*pi reduction (ddspp_common.h); they are chosen per block, wave-uniformly, from
    // conservative bounds on the block's frequencies and phases (block_class below).
    auto process_block = [&](int n0, int tpos, const float (*fb)[VPL], const float (*ab)[VPL],
                             auto fdiv_tag, auto fmod_tag) {
        constexpr bool FDIV = decltype(fdiv_tag)::value;
        constexpr bool FMOD = decltype(fmod_tag)::value;
        float fe[BLK][VPL], ae[BLK][VPL];
        if (FUSED) {
            // per-sample scalar weights (wave-uniform -> scalar loads)
            float wl[BLK], w0[BLK], w1[BLK];
#pragma unroll
            for (int i = 0; i < BLK; ++i) {
                wl[i] = p.wlin[n0 + i];
                w0[i] = p.whann[U + r + i];
                w1[i] = p.whann[r + i];
            }
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const float dx = x1[j] - x0[j];
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    fe[i][j] = x0[j] + dx * wl[i];                    // legacy bilinear (core.resample)
                    // Hann overlap-add of frames t and t+1 (core.upsample_with_windows); the second
                    // product is fused: <= 1 ulp on an amplitude, never on a phase.
                    if (MODE != MODE_PREPASS) ae[i][j] = __builtin_fmaf(a1[j], w1[i], a0[j] * w0[i]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < BLK; ++i)
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    fe[i][j] = fb[i][j];
                    ae[i][j] = ab[i][j];
                }
        }
        // sequential float32 phase scan
        float pv[BLK][VPL];
#pragma unroll
        for (int i = 0; i < BLK; ++i)
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                ph[j] = ph[j] + omega_of<FDIV>(fe[i][j], p.sr, p.rsr);
                pv[i][j] = ph[j];
            }
        if (MODE == MODE_PREPASS) return;
#pragma unroll
        for (int i = 0; i < BLK; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const float a = (fe[i][j] >= p.nyq) ? 0.0f : ae[i][j];   // remove_above_nyquist
                float c;
                if (MODE == MODE_MAIN) {
                    const float s = pv[i][j] + off[j];                    // phase + offsets
                    c = FMOD ? cos_of_phase_fast(s) : cos_reduced(mod_2pi(s));   // % 2pi ; cos
                } else {
                    c = cosf(pv[i][j]);                                   // plain tf.cumsum path
                }
                if (SUM) {
                    acc = __builtin_fmaf(a, c, acc);
                } else if (valid[j]) {
                    p.out[((size_t)row * N + n0 + i) * V + lane + 64 * j] = a * c;
                }
            }
            if (SUM) tile[(tpos + i) * TSTRIDE + lane] = acc;
        }
    };

    auto flush_tile = [&](int nt0, int count) {
        // column sums: lane (col, half) adds 32 of the 64 lane partials of sample `col`
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int col = lane & 31, half = lane >> 5;
        float s = 0.0f;
        const float* src = tile + col * TSTRIDE + half * 32;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += src[i];
        s += __shfl_xor(s, 32);
        if (lane < count) p.out[(size_t)row * N + nt0 + lane] = s;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // 2 = frequencies are all >= 0 and either 0 or comfortably normal, and the phase stays below
    //     2^22 * 2pi through the block: exact fast division (if the rate is whitelisted) + fast mod
    // 0 = anything else (negative / denormal frequencies, huge phases): IEEE divide, generic floormod
    auto block_class = [&](const float (*fb)[VPL]) -> int {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            float lo, hi;
            if (FUSED) {
                lo = fminf(x0[j], x1[j]);
                hi = fmaxf(x0[j], x1[j]);
            } else {
                lo = hi = fb[0][j];
#pragma unroll
                for (int i = 1; i < BLK; ++i) {
                    lo = fminf(lo, fb[i][j]);
                    hi = fmaxf(hi, fb[i][j]);
                }
            }
            ok = ok && (lo == 0.0f || lo > 1e-28f) && (hi == hi) &&
                 (ph[j] + hi * (8.1f * DDSPP_TWO_PI_F32) * p.rsr < 2.5e7f);
            // a block whose lowest frequency is 0 may still hold tiny positive ones: only the fused
            // source can guarantee monotone interpolation between x0 and x1
            if (!FUSED && lo == 0.0f) {
#pragma unroll
                for (int i = 0; i < BLK; ++i) ok = ok && (fb[i][j] == 0.0f || fb[i][j] > 1e-28f);
            }
            if (FUSED && lo == 0.0f) ok = ok && (hi == 0.0f || hi > 1e-24f);
        }
        return __all(ok) ? 2 : 0;
    };

    int tpos = 0;                      // position inside the LDS tile
    int tile_n0 = n_begin;
    if (!FUSED) load_block(n_begin, fbuf[0], abuf[0]);
    int cur = 0;

    for (int n0 = n_begin; n0 < n_end; n0 += BLK) {
        if (!FUSED) {
            // prefetch the next block into the other register buffer
            if (cur == 0) load_block(n0 + BLK, fbuf[1], abuf[1]);
            else load_block(n0 + BLK, fbuf[0], abuf[0]);
        }
        {
            const float (*fbc)[VPL] = cur == 0 ? fbuf[0] : fbuf[1];
            const float (*abc)[VPL] = cur == 0 ? abuf[0] : abuf[1];
            using T_ = std::true_type;
            using F_ = std::false_type;
            const int cls = block_class(fbc);
            if (cur == 0) {
                if (cls == 2 && p.fastdiv) process_block(n0, tpos, fbuf[0], abuf[0], T_{}, T_{});
                else if (cls == 2) process_block(n0, tpos, fbuf[0], abuf[0], F_{}, T_{});
                else process_block(n0, tpos, fbuf[0], abuf[0], F_{}, F_{});
            } else {
                if (cls == 2 && p.fastdiv) process_block(n0, tpos, fbuf[1], abuf[1], T_{}, T_{});
                else if (cls == 2) process_block(n0, tpos, fbuf[1], abuf[1], F_{}, T_{});
                else process_block(n0, tpos, fbuf[1], abuf[1], F_{}, F_{});
            }
            (void)fbc; (void)abc;
        }
        if (!FUSED) cur ^= 1;

        // ---- tile bookkeeping -----------------------------------------------------------------
        if (SUM && MODE != MODE_PREPASS) {
            tpos += BLK;
            if (tpos == TILE || n0 + BLK >= n_end) {
                flush_tile(tile_n0, tpos);
                tile_n0 += tpos;
                tpos = 0;
            }
        }
        // ---- chunk boundary (ddsp.core.angular_cumsum) ------------------------------------------
        if (MODE != MODE_PLAIN) {
            cpos += BLK;
            if (cpos == DDSPP_CHUNK) {
                cpos = 0;
                if (MODE == MODE_PREPASS) {
#pragma unroll
                    for (int j = 0; j < VPL; ++j)
                        p.ework[((size_t)row * p.npre + chunk) * p.VP + lane + 64 * j] = mod_2pi(ph[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < VPL; ++j) {
                        const float e = mod_2pi(ph[j]);      // phase[:, :, -1] % 2pi
                        asum[j] = asum[j] + e;               // cumsum over chunks (float32, sequential)
                        off[j] = mod_2pi(asum[j]);           // % 2pi
                        ph[j] = 0.0f;
                    }
                }
                ++chunk;
            }
        }
        // ---- frame boundary (fused source) -------------------------------------------------------
        if (FUSED) {
            r += BLK;
            if (r == U) {
                r = 0;
                ++t;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    x0[j] = x1[j];
                    a0[j] = a1[j];
                }
                frame_ctl(min(t + 1, T - 1), x1, a1);
            }
        }
    }
}

// Sequential (float32) scan of the chunk end phases: astart[row, span, v] = e[0] + ... + e[c0-1]
// in exactly the order of `tf.cumsum(offsets, axis=1)` in ddsp.core.angular_cumsum.
__global__ void __launch_bounds__(256) osc_offset_scan_kernel(const float* __restrict__ ework,
                                                            float* __restrict__ astart, int R,
                                                            int npre, int VP, int spans, int cps) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)R * VP) return;
    const int row = (int)(gid / VP), v = (int)(gid - (size_t)row * VP);
    float a = 0.0f;
    int span = 0;
    for (int c = 0; c <= npre; ++c) {
        if (c == span * cps && span < spans) {
            astart[((size_t)row * spans + span) * VP + v] = a;
            ++span;
        }
        if (c < npre) a = a + ework[((size_t)row * npre + c) * VP + v];
    }
}

static bool sample_rate_is_checked(float sr) {
    // rates for which div_const == IEEE division was verified for every float32 input with
    // |x| >= 1e-28 (tests/test_exact_arith.py builds and runs the checker)
    const float ok[] = {8000.f, 16000.f, 22050.f, 24000.f, 32000.f, 44100.f, 48000.f, 96000.f};
    for (float v : ok)
        if (sr == v) return true;
    return false;
}

static int env_int(const char* name, int dflt) {
    const char* s = getenv(name);
    if (!s || !*s) return dflt;
    return atoi(s);
}

static int pick_vpl(int V) {
    const int need = (V + 63) / 64;
    const int avail[] = {1, 2, 3, 4, 6, 8};
    for (int a : avail)
        if (a >= need) return a;
    return 0;
}

struct Plan {
    int vpl, VP, nchunks, spans, cps, npre;
    size_t ework_floats, astart_floats;
};

static Plan make_plan(int R, int N, int V, bool angular, bool fused, int spans_req) {
    Plan pl{};
    pl.vpl = pick_vpl(V);
    pl.VP = pl.vpl * 64;
    pl.nchunks = (N + DDSPP_CHUNK - 1) / DDSPP_CHUNK;
    int spans = 1;
    if (angular) {
        if (spans_req > 0) {
            spans = spans_req;
        } else {
            // enough wavefronts to occupy 256 CUs; the fused source is ALU bound and likes more
            // waves, the materialised source pays one extra read of `fe` per pre-passed chunk and
            // is kept at one span whenever the rows alone give >= 4 waves per CU.
            const int target = fused ? env_int("DDSPP_OSC_TARGET_WAVES_FUSED", 4096)
                                     : env_int("DDSPP_OSC_TARGET_WAVES", 1024);
            spans = (target + R - 1) / R;
        }
        if (spans < 1) spans = 1;
        if (spans > pl.nchunks) spans = pl.nchunks;
    }
    pl.cps = (pl.nchunks + spans - 1) / spans;
    pl.spans = (pl.nchunks + pl.cps - 1) / pl.cps;
    pl.npre = pl.spans > 1 ? (pl.spans - 1) * pl.cps : 0;
    pl.ework_floats = (size_t)R * pl.npre * pl.VP;
    pl.astart_floats = (size_t)R * pl.spans * pl.VP;
    return pl;
}

template <int VPL, bool FUSED>
static void launch_all(const OscParams& p, bool angular, bool sum, hipStream_t stream) {
    const int nblk_main = (p.R * p.spans + 3) / 4;
    if (angular) {
        if (p.spans > 1) {
            const int nblk_pre = (p.R * p.npre + 3) / 4;
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_PREPASS, true>), dim3(nblk_pre), dim3(256),
                               0, stream, p);
            const size_t nthr = (size_t)p.R * p.VP;
            hipLaunchKernelGGL(osc_offset_scan_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256),
                               0, stream, p.ework, const_cast<float*>(p.astart), p.R, p.npre, p.VP,
                               p.spans, p.cps);
        }
        if (sum)
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_MAIN, true>), dim3(nblk_main), dim3(256), 0,
                               stream, p);
        else
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_MAIN, false>), dim3(nblk_main), dim3(256),
                               0, stream, p);
    } else {
        if (sum)
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_PLAIN, true>), dim3(nblk_main), dim3(256),
                               0, stream, p);
        else
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_PLAIN, false>), dim3(nblk_main), dim3(256),
                               0, stream, p);
    }
}

template <bool FUSED>
static int dispatch_vpl(int vpl, const OscParams& p, bool angular, bool sum, hipStream_t stream) {
    switch (vpl) {
        case 1: launch_all<1, FUSED>(p, angular, sum, stream); break;
        case 2: launch_all<2, FUSED>(p, angular, sum, stream); break;
        case 3: launch_all<3, FUSED>(p, angular, sum, stream); break;
        case 4: launch_all<4, FUSED>(p, angular, sum, stream); break;
        case 6: launch_all<6, FUSED>(p, angular, sum, stream); break;
        case 8: launch_all<8, FUSED>(p, angular, sum, stream); break;
        default: return DDSPP_EINVAL;
    }
    return DDSPP_OK;
}

}  // namespace ddspp

using namespace ddspp;

extern "C" {

// Workspace (bytes) the two oscillator entry points may need for R rows of N samples and V
// oscillators per row (V = n_substrings * n_harmonics for the fused entry point).
size_t ddspp_osc_workspace_bytes(int R, int N, int V) {
    if (R <= 0 || N <= 0 || V <= 0) return 0;
    const int vpl = pick_vpl(V);
    if (!vpl) return 0;
    const size_t nchunks = (N + DDSPP_CHUNK - 1) / DDSPP_CHUNK;
    return 2 * (size_t)R * nchunks * vpl * 64 * sizeof(float) + 256;
}

// cos_oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate, sum_sinusoids,
// use_angular_cumsum)   -- ddsp_piano/modules/inharm_synth.py:49-84
int ddspp_cos_oscillator_bank(const float* frequency_envelopes, const float* amplitude_envelopes,
                              float* audio, int R, int N, int H, float sample_rate, int sum_sinusoids,
                              int use_angular_cumsum, int spans, void* workspace,
                              size_t workspace_bytes, hipStream_t stream) {
    DDSPP_REQUIRE(frequency_envelopes && amplitude_envelopes && audio, "cos_oscillator_bank: null buffer");
    DDSPP_REQUIRE(R > 0 && N > 0 && H > 0, "cos_oscillator_bank: bad dims R=%d N=%d H=%d", R, N, H);
    DDSPP_REQUIRE(pick_vpl(H) != 0, "cos_oscillator_bank: n_sinusoids=%d exceeds 512", H);
    DDSPP_REQUIRE(N % BLK == 0, "cos_oscillator_bank: n_samples=%d must be a multiple of %d", N, BLK);
    DDSPP_REQUIRE(sample_rate > 0.f, "cos_oscillator_bank: bad sample_rate");
    Plan pl = make_plan(R, N, H, use_angular_cumsum != 0, false,
                        spans > 0 ? spans : env_int("DDSPP_OSC_SPANS", 0));
    const size_t need = (pl.ework_floats + pl.astart_floats) * sizeof(float);
    DDSPP_REQUIRE(pl.spans == 1 || (workspace && workspace_bytes >= need),
                  "cos_oscillator_bank: workspace too small (%zu < %zu)", workspace_bytes, need);
    OscParams p{};
    p.fe = frequency_envelopes;
    p.ae = amplitude_envelopes;
    p.out = audio;
    p.ework = (float*)workspace;
    p.astart = p.ework ? p.ework + pl.ework_floats : nullptr;
    p.R = R; p.N = N; p.T = 0; p.U = BLK; p.H = H; p.S = 1; p.V = H; p.VP = pl.VP;
    p.spans = pl.spans; p.cps = pl.cps; p.nchunks = pl.nchunks; p.npre = pl.npre;
    p.sr = sample_rate; p.rsr = 1.0f / sample_rate; p.nyq = sample_rate / 2.0f;
    p.fastdiv = sample_rate_is_checked(sample_rate) && !env_int("DDSPP_NO_FASTDIV", 0);
    int rc = dispatch_vpl<false>(pl.vpl, p, use_angular_cumsum != 0, sum_sinusoids != 0, stream);
    DDSPP_REQUIRE(rc == DDSPP_OK, "cos_oscillator_bank: dispatch failed");
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// harmonic_synthesis(...) summed over the substrings of MultiInharmonic.get_signal, straight from
// the frame-rate controls -- inharm_synth.py:87-127, :272-293.  `wlin` and `whann` come from
// ddspp_resample_tables (resample.hip); the [R, N, H] envelopes are never materialised.
int ddspp_harmonic_synthesis(const float* f0_hz, const float* amplitudes,
                             const float* harmonic_distribution, const float* harmonic_shifts,
                             const float* wlin, const float* whann, float* audio, int R, int T, int S,
                             int H, int U, float sample_rate, int use_angular_cumsum, int spans,
                             void* workspace, size_t workspace_bytes, hipStream_t stream) {
    DDSPP_REQUIRE(f0_hz && amplitudes && harmonic_distribution && wlin && whann && audio,
                  "harmonic_synthesis: null buffer");
    DDSPP_REQUIRE(R > 0 && T > 0 && S > 0 && H > 0 && U > 0, "harmonic_synthesis: bad dims");
    DDSPP_REQUIRE(U % BLK == 0, "harmonic_synthesis: upsampling=%d must be a multiple of %d "
                  "(use the resample + cos_oscillator_bank route)", U, BLK);
    const int V = S * H;
    DDSPP_REQUIRE(pick_vpl(V) != 0, "harmonic_synthesis: n_substrings*n_harmonics=%d exceeds 512", V);
    DDSPP_REQUIRE((long long)T * U < (1ll << 31), "harmonic_synthesis: too many samples");
    const int N = T * U;
    Plan pl = make_plan(R, N, V, use_angular_cumsum != 0, true,
                        spans > 0 ? spans : env_int("DDSPP_OSC_SPANS_FUSED", 0));
    const size_t need = (pl.ework_floats + pl.astart_floats) * sizeof(float);
    DDSPP_REQUIRE(pl.spans == 1 || (workspace && workspace_bytes >= need),
                  "harmonic_synthesis: workspace too small (%zu < %zu)", workspace_bytes, need);
    OscParams p{};
    p.f0 = f0_hz; p.amp = amplitudes; p.hd = harmonic_distribution; p.shifts = harmonic_shifts;
    p.wlin = wlin; p.whann = whann;
    p.out = audio;
    p.ework = (float*)workspace;
    p.astart = p.ework ? p.ework + pl.ework_floats : nullptr;
    p.R = R; p.N = N; p.T = T; p.U = U; p.H = H; p.S = S; p.V = V; p.VP = pl.VP;
    p.spans = pl.spans; p.cps = pl.cps; p.nchunks = pl.nchunks; p.npre = pl.npre;
    p.sr = sample_rate; p.rsr = 1.0f / sample_rate; p.nyq = sample_rate / 2.0f;
    p.fastdiv = sample_rate_is_checked(sample_rate) && !env_int("DDSPP_NO_FASTDIV", 0);
    int rc = dispatch_vpl<true>(pl.vpl, p, use_angular_cumsum != 0, true, stream);
    DDSPP_REQUIRE(rc == DDSPP_OK, "harmonic_synthesis: dispatch failed");
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

}  // extern "C"
