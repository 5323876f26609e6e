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
//   (the compacted polyphonic bank -- what the batched group runs -- lives in bank_compact.hip)
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

#include <atomic>

#include "osc_common.h"

// One source, four translation units (round 6: the template instances of osc_kernel / osc_prepass_fused_kernel were 1 min 55 s
// of a 2-minute cold build in ONE compiler process): this file compiled as it is (part 0) holds every non-template kernel,
// the host side, the C-ABI and the instances for 1 and 2 oscillators per lane; oscillator_p1/2/3.hip define DDSPP_OSC_PART
// and include it again for the instances with 3-4, 6 and 8 oscillators per lane (rows of 129 .. 512 oscillators: S x H of the
// fused entry points, H of cos_oscillator_bank -- every shipped configuration except ENSTDkCl-32kHz (192) and the two-string
// 24 kHz one (256) stays within part 0).  The parts build in parallel (ddsp_piano_amd/_lib.py).
#ifndef DDSPP_OSC_PART
#define DDSPP_OSC_PART 0
#endif
#define DDSPP_OSC_OWNS(vpl)                                                                                     \
    ((DDSPP_OSC_PART == 0 && (vpl) <= 2) || (DDSPP_OSC_PART == 1 && ((vpl) == 3 || (vpl) == 4)) ||              \
     (DDSPP_OSC_PART == 2 && (vpl) == 6) || (DDSPP_OSC_PART == 3 && (vpl) == 8))

namespace ddspp {

// helpers of the launchers below, defined in part 0
int env_int(const char* name, int dflt);
void launch_offset_scan(const float* ework, float* astart, int R, int npre, int VP, int spans, int cps, hipStream_t stream,
                        const float* state_in = nullptr, int V = 0);
void launch_memo_prepass(int vpl, const OscParams& q0, int tasks, float* echunk, hipStream_t stream);

// DECAY (fused source, SurrogateAdditive): the amplitude of oscillator k in frame t is multiplied by
// |decays[t, k]| ** (decay_time[t] U + r), r = sample in the frame (surrogate_synth.py:76-95: tf.repeat of the frame's
// values, a sample counter, tf.math.pow).  The power is evaluated by powf once per frame and lane (and where a span
// starts inside a frame); inside the frame it advances block by block with d ** 8 and sample by sample with d: at most
// U / 8 + 10 roundings behind the per-sample pow of the reference (2e-6 relative at hop 192), on an amplitude only.
template <int VPL, bool FUSED, int MODE, bool SUM, bool DECAY = false>
__global__ void __launch_bounds__(256) osc_kernel(const OscParams p) {
    static_assert(!DECAY || (FUSED && MODE != MODE_PREPASS && SUM), "the decay term belongs to the fused, summed source");
    // one workgroup = the `groups` wavefronts of one (row, span): they walk the same samples, so
    // their per-tile partial sums can be combined through LDS behind a single barrier per tile
    extern __shared__ float lds_dyn[];

    const int lane = threadIdx.x & 63;
    const int wib = wave_uniform(threadIdx.x >> 6);
    const int grp = wib;
    const int task = blockIdx.x;
    int row, c0, c1;
    if (MODE == MODE_PREPASS) {
        row = task / p.npre;
        c0 = task - row * p.npre;
        c1 = c0 + 1;
    } else {
        row = task / p.spans;
        const int span = task - row * p.spans;
        c0 = span * p.cps;
        c1 = min(c0 + p.cps, p.nchunks);
    }
    const int vbase = grp * p.vgrp;                       // first oscillator of this wavefront
    const int vlast = min(vbase + p.vgrp, p.V) - 1;       // last one (inclusive)
    float* tile = lds_dyn + grp * (TILE * TSTRIDE);
    float* comb = lds_dyn + p.groups * (TILE * TSTRIDE);      // [2][groups][32] combine buffer
    int comb_buf = 0;

    const int N = p.N, U = p.U, H = p.H, T = p.T, S = p.S, V = p.V;
    // wave-uniform read-only tables: the constant address space makes the compiler fetch them with
    // scalar loads (s_load_dwordx8 -> SGPR operands) instead of per-lane vector loads
    typedef const __attribute__((address_space(4))) float* cfloat_p;
    const cfloat_p wlin_c = (cfloat_p)(uintptr_t)p.wlin;
    const cfloat_p whann_c = (cfloat_p)(uintptr_t)p.whann;
    const int n_begin = c0 * DDSPP_CHUNK;
    const int n_end = min(c1 * DDSPP_CHUNK, N);
    const float nyq = p.nyq;

    // ---- per-lane oscillator identity.  Lanes past the last oscillator load a clamped (valid)
    // address and are silenced arithmetically: no load ever sits behind a branch, so the compiler
    // keeps counted vmcnt waits and the prefetched blocks really stay in flight.
    // Oscillator index of (lane, j): the fused source keeps harmonics 64 j .. 64 j + 63 together in group
    // j (the upper groups go silent first, see act[]); the materialised source gives each lane VPL
    // adjacent sinusoids so that one 4/8/16-byte load per lane fetches them (the host picks such a
    // VPL only when H % VPL == 0).
    constexpr bool CONTIG = !FUSED && (VPL == 1 || VPL == 2 || VPL == 4);
    int vk[VPL], vs[VPL], vcol[VPL], vidx[VPL];
    bool valid[VPL];
    float kmul[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int v = vbase + (CONTIG ? lane * VPL + j : lane + 64 * j);
        vidx[j] = v;
        valid[j] = v <= vlast;
        const int vc = (CONTIG && VPL > 1) ? min(vbase + lane * VPL, vlast + 1 - VPL) + j : min(v, vlast);
        vs[j] = vc / H;
        vk[j] = vc - vs[j] * H;
        vcol[j] = vc;
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
            asum[j] = p.astart[((size_t)row * p.spans + span) * p.VP + vidx[j]];
            off[j] = chunk_offset(asum[j], p.off_plain);
        }
    }

    // ---- fused source: frame controls ----------------------------------------------------------
    // hf(t, v) = (f0[t, s] * k) * (1 + shift[t, k])     inharm_synth.py:106-108
    // ha(t, v) = amp[t] * hd[t, k]                      inharm_synth.py:112
    // x0/a0 = frame t, x1/a1 = frame min(t + 1, T - 1); the raw values of the frame after that are
    // requested one whole frame early (q_*), so their HBM/L2 latency hides behind 96 samples of work.
    float x0[VPL], x1[VPL], a0[VPL], a1[VPL];
    float a1_raw[VPL];                 // frame t + 1's amplitude before classify_frame's whole-pair Nyquist mask: the NEXT pair starts from it
    float q_f0[VPL], q_sh[VPL], q_hd[VPL], q_amp[VPL];
    // DECAY: d0 / dt0 = frame t's decay factor and time, d1 / dt1 = frame t + 1's, q_d / q_dt those of the frame
    // requested ahead; e_blk = d0 ** (dt0 U + r) at the start of the current block, d0_8 = d0 ** 8
    float d0[DECAY ? VPL : 1], d1[DECAY ? VPL : 1], q_d[DECAY ? VPL : 1], e_blk[DECAY ? VPL : 1], d0_8[DECAY ? VPL : 1];
    float dt0 = 0.0f, dt1 = 0.0f, q_dt = 0.0f;
    int t = 0, r = 0;
    auto frame_request = [&](int tt) {
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const size_t fr = (size_t)row * T + tt;
            if constexpr (DECAY) {
                q_d[j] = fabsf(p.decays[fr * H + vk[j]]);
                q_dt = p.decay_time[fr];
            }
            q_amp[j] = p.amp[fr];
            q_f0[j] = p.f0[fr * S + vs[j]];
            q_sh[j] = p.shifts ? p.shifts[fr * H + vk[j]] : (p.inh ? p.inh[fr] : 0.0f);
            q_hd[j] = (MODE != MODE_PREPASS) ? p.hd[fr * H + vk[j]] : 0.0f;
        }
    };
    auto frame_finish = [&](float* xf, float* xa) {
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            float f = q_f0[j] * kmul[j];
            if (p.shifts) f = f * (1.0f + q_sh[j]);
            else if (p.inh) f = f * (1.0f + shift_from_inharm(q_sh[j], kmul[j]));
            const float a = q_amp[j] * q_hd[j];
            xf[j] = valid[j] ? f : 0.0f;
            xa[j] = (valid[j] && MODE != MODE_PREPASS) ? a : 0.0f;
        }
    };

    // ---- per-frame / per-block classification (wave-uniform) -----------------------------------
    //   vals_ok   every frequency is >= 0 and either 0 or comfortably normal  -> exact fast forms of
    //             the constant division and of the 2*pi reduction are valid (ddspp_common.h)
    //   need_mask some oscillator crosses Nyquist inside the frame pair        -> per-sample mask
    //   act[j]    some lane of group j has a non-zero amplitude                -> otherwise the group
    //             contributes exactly 0 and only its phase is advanced
    bool vals_ok = false, need_mask = true, const_freq = false;
    bool act[VPL];
    float phlim[VPL], om_c[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        act[j] = true;
        phlim[j] = 0.0f;
        om_c[j] = 0.0f;
    }
    auto classify_frame = [&]() {
        bool ok = true, msk = false, cst = true;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const float lo = fminf(x0[j], x1[j]), hi = fmaxf(x0[j], x1[j]);
            ok = ok && (lo > 1e-28f || (lo == 0.0f && (hi == 0.0f || hi > 1e-24f))) && (hi < 3.0e38f);
            if (lo >= nyq) {            // above Nyquist for the whole frame pair: mask once, here
                a0[j] = 0.0f;
                a1[j] = 0.0f;
            }
            msk = msk || (lo < nyq && hi >= nyq);
            cst = cst && (x0[j] == x1[j]);
            om_c[j] = omega_of<false>(x0[j], p.sr, p.rsr);   // == omega of every sample when x0 == x1
            phlim[j] = 2.5e7f - hi * (8.1f * DDSPP_TWO_PI_F32) * p.rsr;
            act[j] = __any(a0[j] != 0.0f || a1[j] != 0.0f);
        }
        vals_ok = __all(ok);
        need_mask = __any(msk);
        const_freq = __all(cst);
    };

    if (FUSED) {
        t = n_begin / U;
        r = n_begin - t * U;
        frame_request(t);
        frame_finish(x0, a0);
        frame_request(min(t + 1, T - 1));
        frame_finish(x1, a1);
#pragma unroll
        for (int j = 0; j < VPL; ++j) a1_raw[j] = a1[j];
        if constexpr (DECAY) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) d1[j] = q_d[j];
            dt1 = q_dt;
        }
        frame_request(min(t + 2, T - 1));
        classify_frame();
    }
    // DECAY: the power at sample r0 of the frame whose factors are in d0 / dt0
    auto decay_start = [&](int r0) {
        if constexpr (DECAY) {
            const float x = dt0 * (float)U + (float)r0;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                e_blk[j] = powf(d0[j], x);
                const float d2 = d0[j] * d0[j], d4 = d2 * d2;
                d0_8[j] = d4 * d4;
            }
        }
    };
    if constexpr (FUSED && DECAY) {
        // frame t's own factors (the requests above were for frames t, t + 1 and t + 2)
#pragma unroll
        for (int j = 0; j < VPL; ++j) d0[j] = fabsf(p.decays[((size_t)row * T + t) * H + vk[j]]);
        dt0 = p.decay_time[(size_t)row * T + t];
        decay_start(r);
    }

    float* out_row = p.out + (size_t)row * N;
    int cpos = 0;                      // position inside the current 1000-sample chunk
    int chunk = c0;
    const float* fe_row = FUSED ? nullptr : p.fe + (size_t)row * N * H;
    const float* ae_row = FUSED ? nullptr : p.ae + (size_t)row * N * H;

    // materialised source: ring of register buffers, NBUF - 1 blocks (8 KB each at H = 128) are in
    // flight per wavefront while one is being consumed.
#ifndef DDSPP_NBUF
#define DDSPP_NBUF 4
#endif
    constexpr int NBUF = FUSED ? 1 : (VPL <= 2 ? DDSPP_NBUF : (VPL <= 4 ? 2 : 1));
    float fbuf[NBUF][BLK][VPL], abuf[NBUF][BLK][VPL];
    auto load_block = [&](int n0, float (*fb)[VPL], float (*ab)[VPL]) {
#pragma unroll
        for (int i = 0; i < BLK; ++i) {
            const size_t base = (size_t)min(n0 + i, n_end - 1) * H;       // clamped: never out of the span
            if (CONTIG && VPL > 1) {
                typedef float vecf __attribute__((ext_vector_type(VPL)));
                const size_t o = base + vcol[0];
                const vecf f = __builtin_nontemporal_load(reinterpret_cast<const vecf*>(fe_row + o));
                vecf a = {};
                if (MODE != MODE_PREPASS) a = __builtin_nontemporal_load(reinterpret_cast<const vecf*>(ae_row + o));
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    fb[i][j] = f[j];
                    ab[i][j] = a[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    fb[i][j] = fe_row[base + vcol[j]];
                    ab[i][j] = (MODE != MODE_PREPASS) ? ae_row[base + vcol[j]] : 0.0f;
                }
            }
        }
    };

    // One block of BLK samples.
    auto process_block = [&](int n0, int tpos, const float (*fb)[VPL], const float (*ab)[VPL],
                             auto fdiv_tag, auto fmod_tag, auto mask_tag, auto cfreq_tag) {
        constexpr bool FDIV = decltype(fdiv_tag)::value;
        constexpr bool FMOD = decltype(fmod_tag)::value;
        constexpr bool MASK = decltype(mask_tag)::value;
        // CFREQ: every oscillator keeps its frequency through the frame pair (x0 == x1, the normal
        // state of a held piano note): fe == x0 exactly, so omega is the per-frame constant om_c.
        constexpr bool CFREQ = decltype(cfreq_tag)::value;
        float fe[BLK][VPL], ae[BLK][VPL];
        if (FUSED) {
            // per-sample scalar weights (wave-uniform -> scalar loads)
            float wl[BLK], w1[BLK];
#pragma unroll
            for (int i = 0; i < BLK; ++i) {
                if (!CFREQ) wl[i] = wlin_c[n0 + i];
                w1[i] = whann_c[r + i];
            }
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const float dx = x1[j] - x0[j];
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    if (!CFREQ) fe[i][j] = x0[j] + dx * wl[i];        // legacy bilinear (core.resample)
                    // Hann overlap-add of frames t and t+1 (core.upsample_with_windows):
                    // a0 w[U + r] + a1 w[r] with w[U + r] + w[r] = 1 (to 1 ulp) is the cross-fade
                    // a0 + (a1 - a0) w[r]; <= 2 ulp on an amplitude, never on a phase.
                    if (MODE != MODE_PREPASS && act[j]) ae[i][j] = __builtin_fmaf(a1[j] - a0[j], w1[i], a0[j]);
                }
                if constexpr (DECAY) {     // amplitude_envelopes *= |decays| ** (decay_time U + r)   surrogate_synth.py:91-95
                    float e = e_blk[j];
#pragma unroll
                    for (int i = 0; i < BLK; ++i) {
                        if (act[j]) ae[i][j] = ae[i][j] * e;
                        e = e * d0[j];
                    }
                    e_blk[j] = e_blk[j] * d0_8[j];
                }
            }
            if (!CFREQ) {
                if (r + BLK == U && wl[BLK - 1] == WALK_NEXT_ROW) {       // a long file: see osc_common.h (wave-uniform)
#pragma unroll
                    for (int i = 0; i < BLK; ++i)
#pragma unroll
                        for (int j = 0; j < VPL; ++j) fe[i][j] = (wl[i] == WALK_NEXT_ROW) ? x1[j] : fe[i][j];
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < BLK; ++i)
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    fe[i][j] = valid[j] ? fb[i][j] : 0.0f;
                    ae[i][j] = valid[j] ? ab[i][j] : 0.0f;
                }
        }
        // sequential float32 phase scan
        float pv[BLK][VPL];
#pragma unroll
        for (int i = 0; i < BLK; ++i)
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                ph[j] = ph[j] + (CFREQ ? om_c[j] : omega_of<FDIV>(fe[i][j], p.sr, p.rsr));
                pv[i][j] = ph[j];
            }
        if (MODE == MODE_PREPASS) return;
        float acc[BLK];
#pragma unroll
        for (int i = 0; i < BLK; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            if (act[j]) {                  // wave-uniform
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    const float a = (MASK && !CFREQ && fe[i][j] >= nyq) ? 0.0f : ae[i][j];   // remove_above_nyquist
                    float c;
                    if (MODE == MODE_MAIN) {
                        const float s = pv[i][j] + off[j];                        // phase + offsets
                        c = FMOD ? cos_of_phase_fast(s) : cos_reduced(mod_2pi(s));   // % 2pi ; cos
                    } else {
                        c = cosf(pv[i][j]);                                       // plain tf.cumsum path
                    }
                    if (SUM) acc[i] = __builtin_fmaf(a, c, acc[i]);
                    else if (valid[j]) p.out[((size_t)row * N + n0 + i) * V + vidx[j]] = a * c;
                }
            } else if (!SUM) {
#pragma unroll
                for (int i = 0; i < BLK; ++i)
                    if (valid[j]) p.out[((size_t)row * N + n0 + i) * V + vidx[j]] = 0.0f;
            }
        }
        if (SUM) {
#pragma unroll
            for (int i = 0; i < BLK; ++i) tile[(tpos + i) * TSTRIDE + lane] = acc[i];
        }
    };

    auto flush_tile = [&](int nt0, int count) {
        // column sums: lane (col, half) adds 32 of the 64 lane partials of sample `col`
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int col = lane & 31, half = lane >> 5;
        // eight 16-byte reads, four running sums (a single chain of 32 dependent adds costs ~5 cycles per add)
        const float4* src = reinterpret_cast<const float4*>(tile + col * TSTRIDE + half * 32);
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v tv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) tv[i] = *reinterpret_cast<const f4v*>(src + i);
        // all eight reads in flight before the first add (the per-block arrays are dead here, the registers are free)
        asm volatile("" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tv[4]), "+v"(tv[5]), "+v"(tv[6]), "+v"(tv[7]));
        float4 s4 = make_float4(tv[0].x, tv[0].y, tv[0].z, tv[0].w);
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            s4.x += tv[i].x; s4.y += tv[i].y; s4.z += tv[i].z; s4.w += tv[i].w;
        }
        float s = (s4.x + s4.y) + (s4.z + s4.w);
        s += __shfl_xor(s, 32);
        if (p.groups == 1) {
            if (lane < count) out_row[nt0 + lane] = s;
        } else {
            float* cb = comb + comb_buf * (p.groups * 32);
            if (lane < 32) cb[grp * 32 + lane] = s;
#if !(defined(DDSPP_OSC_ABLATE) && (DDSPP_OSC_ABLATE & 2))   // timing only: the wavefronts of a row never meet
            __syncthreads();
#endif
            if (grp == 0 && lane < count) {
                float tot = cb[lane];
                for (int g = 1; g < p.groups; ++g) tot += cb[g * 32 + lane];
                out_row[nt0 + lane] = tot;
            }
            comb_buf ^= 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // materialised source: the same classification, from the block's own envelope samples
    auto classify_block = [&](const float (*fb)[VPL], const float (*ab)[VPL]) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            float lo = fb[0][j], hi = fb[0][j], amax = fabsf(ab[0][j]);
            bool tiny = false;
#pragma unroll
            for (int i = 0; i < BLK; ++i) {
                lo = fminf(lo, fb[i][j]);
                hi = fmaxf(hi, fb[i][j]);
                amax = fmaxf(amax, fabsf(ab[i][j]));
                tiny = tiny || (fb[i][j] != 0.0f && fb[i][j] < 1e-28f);
            }
            ok = ok && (lo >= 0.0f) && !tiny && (hi < 3.0e38f);
            phlim[j] = 2.5e7f - hi * (8.1f * DDSPP_TWO_PI_F32) * p.rsr;
            act[j] = __any(valid[j] && !(amax == 0.0f));
        }
        vals_ok = __all(ok || !valid[0]) && __all(ok);
        need_mask = true;
    };

    int tpos = 0;                      // position inside the LDS tile
    int tile_n0 = n_begin;

    // everything that happens for one block of BLK samples starting at n0, whose envelopes (for the
    // materialised source) sit in fb / ab
    auto do_block = [&](int n0, const float (*fb)[VPL], const float (*ab)[VPL]) {
        using T_ = std::true_type;
        using F_ = std::false_type;
#if defined(DDSPP_OSC_ABLATE) && (DDSPP_OSC_ABLATE & 1)      // timing only: the loads and nothing else
        if (!FUSED) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < BLK; ++i)
#pragma unroll
                for (int j = 0; j < VPL; ++j) s += fb[i][j] + ab[i][j];
            if (s == 1.2345e30f) p.out[0] = s;
            return;
        }
#endif
        if (!FUSED) classify_block(fb, ab);
        bool fast = vals_ok;
        if (MODE != MODE_PLAIN) {
            bool ph_ok = true;
#pragma unroll
            for (int j = 0; j < VPL; ++j) ph_ok = ph_ok && (ph[j] < phlim[j]);
            fast = fast && __all(ph_ok);
        }
        if (FUSED && fast && const_freq) {
            process_block(n0, tpos, fb, ab, F_{}, T_{}, F_{}, T_{});
        } else if (fast && p.fastdiv) {
            if (FUSED && !need_mask) process_block(n0, tpos, fb, ab, T_{}, T_{}, F_{}, F_{});
            else process_block(n0, tpos, fb, ab, T_{}, T_{}, T_{}, F_{});
        } else if (fast) {
            process_block(n0, tpos, fb, ab, F_{}, T_{}, T_{}, F_{});
        } else {
            process_block(n0, tpos, fb, ab, F_{}, F_{}, T_{}, F_{});
        }
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
                        p.ework[((size_t)row * p.npre + chunk) * p.VP + vidx[j]] = mod_2pi(ph[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < VPL; ++j) {
                        const float e = mod_2pi(ph[j]);      // phase[:, :, -1] % 2pi
                        asum[j] = asum[j] + e;               // cumsum over chunks (float32, sequential)
                        off[j] = chunk_offset(asum[j], p.off_plain);           // % 2pi
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
                    a0[j] = a1_raw[j];
                }
                frame_finish(x1, a1);                       // raw values requested one frame ago
#pragma unroll
                for (int j = 0; j < VPL; ++j) a1_raw[j] = a1[j];
                if constexpr (DECAY) {
#pragma unroll
                    for (int j = 0; j < VPL; ++j) {
                        d0[j] = d1[j];
                        d1[j] = q_d[j];
                    }
                    dt0 = dt1;
                    dt1 = q_dt;
                    decay_start(0);
                }
                frame_request(min(t + 2, T - 1));
                classify_frame();
            }
        }
    };

    if (FUSED) {
        for (int n0 = n_begin; n0 < n_end; n0 += BLK) do_block(n0, fbuf[0], abuf[0]);
    } else {
#pragma unroll
        for (int b = 0; b < NBUF - 1; ++b) load_block(n_begin + b * BLK, fbuf[b], abuf[b]);
        // whole turns of the ring without a branch around a load: hipcc's s_waitcnt bookkeeping joins the states of all
        // paths into a block, and with `if (nb0 < n_end)` around every ring position (rounds 1-5) the waits assumed the
        // skipped loads had never been issued -- vmcnt(16 .. 47) where 48 .. 62 is right, one or two blocks in flight
        // instead of three (found in round 6, osc_stream.hip)
        int n0 = n_begin;
        for (; n0 + NBUF * BLK <= n_end; n0 += NBUF * BLK) {
#pragma unroll
            for (int b = 0; b < NBUF; ++b) {
                load_block(n0 + (b + NBUF - 1) * BLK, fbuf[(b + NBUF - 1) % NBUF], abuf[(b + NBUF - 1) % NBUF]);
                do_block(n0 + b * BLK, fbuf[b], abuf[b]);
            }
        }
        // the last, partial turn (at most NBUF - 1 blocks): they are in the ring already
#pragma unroll
        for (int b = 0; b < NBUF - 1; ++b)
            if (n0 + b * BLK < n_end) do_block(n0 + b * BLK, fbuf[b], abuf[b]);
    }
}

// Fused source, spans > 1: chunk end phases and span start offsets in ONE sequential walk per
// (row, 64-oscillator group), with memoisation.  A chunk whose oscillators all keep one frequency
// (every frame it touches has the same hf -- a held piano note) has an end phase that depends on that
// frequency only (1000 sequential float32 adds of the same omega from 0), so it is scanned once and
// re-used for every later chunk with the same frequencies; only chunks with moving frequencies are
// scanned sample by sample.  Output: astart[row, span, v] = e[0] + ... + e[c0(span) - 1] accumulated
// sequentially in float32 -- exactly what the pre-pass + offset-scan kernels produce.
//
// PARTS > 1: a workgroup is one SECTION of a (row, group): its PARTS wavefronts each walk a short contiguous run of the
// row's chunks (own memo, own change detector started at the run's first frame) and leave the end phases e[c] in
// p.echunk [R, npre, VP]; osc_offset_scan_groups_kernel then adds them up in chunk order -- the same float32 sums in the
// same order.  Why sections: a (row, group) whose frequencies move in every frame (vibrato, glides) costs 72
// sample-by-sample chunk scans, one whose notes are held or silent nearly nothing, and which is which is the input's
// business: with one workgroup per (row, group) (round 2) all 2048 workgroups were resident at once, a CU that drew seven
// moving ones worked 1.8x the average and the chip waited for it (VALU issue fraction 0.43).  Five sections are 10240
// workgroups of a sixth of the length, handed out as earlier ones finish.
// max over the frames of a row of the per-frame audible-harmonic counts (low 16 bits of `audible`), twelve loads in
// flight per lane: a wavefront that starts with this pays one memory latency for a 3 s row, not twelve
__device__ __forceinline__ int row_audible_max(const int* __restrict__ aud, int T, int lane) {
    int amax = 0;
    for (int t0 = 0; t0 < T; t0 += 768) {
        int a[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) {
            const int tt = t0 + u * 64 + lane;
            a[u] = aud[min(tt, T - 1)];
        }
#pragma unroll
        for (int u = 0; u < 12; ++u) amax = max(amax, a[u] & 0xffff);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = max(amax, __shfl_xor(amax, o));
    return amax;
}

template <int VPL, int PARTS>
__global__ void __launch_bounds__(256) osc_prepass_fused_kernel(const OscParams p) {
    extern __shared__ float lds_dyn[];
    const int lane = threadIdx.x & 63;
    const int part = wave_uniform(threadIdx.x >> 6);
    const int unit = PARTS > 1 ? (int)blockIdx.x : wave_uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (PARTS == 1 && unit >= p.R * p.groups) return;
    // group-major task order: consecutive workgroups (which go round-robin to the 8 XCDs) are consecutive ROWS.  With the
    // groups of a row next to each other, group 1 of 2 (partials 65..128: above Nyquist for most notes, no work) took
    // every odd XCD and the four even ones did all the scanning.  PARTS > 1: unit = (group, section, row).
    const int nsec = PARTS > 1 ? p.nsec : 1;
    const int gs = unit / p.R, row = unit - gs * p.R;
    const int grp = gs / nsec, sec = gs - grp * nsec;
    const int T = p.T, U = p.U, H = p.H, S = p.S, N = p.N;
    const int vbase = grp * p.vgrp, vlast = min(vbase + p.vgrp, p.V) - 1;
    // LDS: one PRE_W-float weight buffer per wavefront
    float* wlds = lds_dyn + (size_t)(threadIdx.x >> 6) * PRE_W;

    int vk[VPL], vs[VPL], vidx[VPL];
    bool valid[VPL];
    float kmul[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int v = vbase + lane + 64 * j;
        vidx[j] = v;
        valid[j] = v <= vlast;
        const int vc = min(v, vlast);
        vs[j] = vc / H;
        vk[j] = vc - vs[j] * H;
        kmul[j] = (float)(vk[j] + 1);
    }
    // Harmonics that are silent in every frame of the row (above Nyquist for this note, or a silent voice) never
    // get a lane in the compacted oscillator bank, so their start phases are not needed: a task made only of such
    // harmonics has nothing to do (all wavefronts of the workgroup share the task: a uniform exit), and in a mixed
    // task the silent 64-lane groups read frame 0 over and over (cache hits) instead of streaming their [T, 64]
    // slice.  Not so for a carried phase state (need_all): the reference advances every partial's phase, heard or not.
    bool need[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) need[j] = true;
    if (p.audible && !p.need_all) {
        const int amax = p.rowmax ? p.rowmax[row] : row_audible_max(p.audible + (size_t)row * T, T, lane);
        bool any_needed = false;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            need[j] = __any(valid[j] && vk[j] < amax);
            any_needed = any_needed || need[j];
        }
        if (!any_needed) return;
    }
    // raw loads and arithmetic are kept apart (and free of branches) so that a batch of frames is
    // fetched with all its loads in flight at once
    // shifts: read from the [R, T, H] tensor, or formed from the row's inharm_coef (p.shifts == null, p.inh given);
    // neither: any finite buffer, ignored
    const bool from_inh = !p.shifts && p.inh;
    const float* shp = p.shifts ? p.shifts : (from_inh ? p.inh : p.hd);
    const int sh_stride = from_inh ? 0 : H;                  // inharm_coef is one value per frame
    auto hf_raw = [&](int tt, float* rf, float* rs) {
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const size_t fr = (size_t)row * T + (need[j] ? tt : 0);
            rf[j] = p.f0[fr * S + vs[j]];
            rs[j] = from_inh ? shp[fr] : shp[fr * H + vk[j]];
        }
    };
    auto hf_calc = [&](const float* rf, const float* rs, float* xf) {
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            float f = rf[j] * kmul[j];
            if (p.shifts) f = f * (1.0f + rs[j]);
            else if (from_inh) f = f * (1.0f + shift_from_inharm(rs[j], kmul[j]));
            xf[j] = valid[j] ? f : 0.0f;
        }
    };
    (void)sh_stride;
    auto hf_of = [&](int tt, float* xf) {
        float rf[VPL], rs[VPL];
        hf_raw(tt, rf, rs);
        hf_calc(rf, rs, xf);
    };

    // Streaming change detector over the frames: last_change = largest t <= t_checked with
    // hf(t) != hf(t - 1) in some lane.  Frames are fetched 32 at a time with independent loads, so
    // a wavefront pays one memory latency per 32 frames, not per frame.
    constexpr int FB = PARTS > 1 ? 8 : (VPL <= 2 ? 32 : (VPL <= 4 ? 16 : 8));     // (a section's run is a few dozen frames)
    const bool by_flags = p.audible && !p.dbg_noflags;
    float x_prev[VPL];
    if (!by_flags) hf_of(0, x_prev);
    int t_checked = 0, last_change = 0;
    // With the per-frame flags of the get_controls kernel (bit 16 of p.audible: frequencies may have moved) the
    // detector reads one int per frame, 64 frames per load, instead of streaming the [T, V] controls.
    int fl_base = 1;                   // first frame of the loaded batch of flags
    unsigned long long fl_mask = 0;    // bit i: frame fl_base + i moved
    int fl_before = 0;                 // last moved frame before fl_base
    bool fl_loaded = false;
    auto flagged_upto = [&](int t_need) {      // last_change := largest t <= t_need whose frequencies moved (0: none)
        while (!fl_loaded || t_need >= fl_base + 64) {
            if (fl_loaded) {
                if (fl_mask) fl_before = fl_base + 63 - __builtin_clzll(fl_mask);
                fl_base += 64;
            }
            const int tf = fl_base + lane;
            const int v = tf <= T - 1 ? p.audible[(size_t)row * T + tf] : 0;
            fl_mask = __ballot((v >> 16) & 1);
            fl_loaded = true;
        }
        const int nbits = t_need - fl_base + 1;                       // frames fl_base .. t_need of the batch
        const unsigned long long m = nbits >= 64 ? fl_mask : (nbits <= 0 ? 0ull : (fl_mask & ((1ull << nbits) - 1)));
        last_change = m ? fl_base + 63 - __builtin_clzll(m) : fl_before;
        t_checked = max(t_checked, t_need);
    };
    auto check_upto = [&](int t_need) {
        if (by_flags) {
            flagged_upto(t_need);
            return;
        }
        while (t_checked < t_need) {
            float rf[FB][VPL], rs[FB][VPL];
#pragma unroll
            for (int u = 0; u < FB; ++u) hf_raw(min(t_checked + 1 + u, T - 1), rf[u], rs[u]);
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                float xb[1][VPL];
                hf_calc(rf[u], rs[u], xb[0]);
                bool ch = false;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    ch = ch || (xb[0][j] != x_prev[j]);
                    x_prev[j] = xb[0][j];
                }
                if (__any(ch) && t_checked + 1 + u <= T - 1) last_change = t_checked + 1 + u;
            }
            t_checked = min(t_checked + FB, T - 1);
            if (t_checked == T - 1) break;
        }
    };

    float asum[VPL], e_memo[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        // streaming: the row continues a signal whose chunk offsets have already summed up to state_in
        asum[j] = (p.state_in && vidx[j] < p.V) ? p.state_in[(size_t)row * p.V + vidx[j]] : 0.0f;
        e_memo[j] = 0.0f;
    }
    int memo_t = -1;                   // first frame of the chunk the memo was taken from (-1: none)
    // this wavefront's run of chunks
    const int per_part = (p.npre + nsec * PARTS - 1) / (nsec * PARTS);
    const int c_begin = PARTS > 1 ? min((sec * PARTS + part) * per_part, p.npre) : 0;
    const int c_end = PARTS > 1 ? min(c_begin + per_part, p.npre) : p.npre;
    if (PARTS > 1 && c_begin > 0) {    // the change detector starts at this run's first frame
        const int t0 = (c_begin * DDSPP_CHUNK) / U;
        t_checked = t0;
        last_change = t0;
        fl_base = t0 + 1;
        fl_before = t0;
        if (!by_flags) hf_of(t0, x_prev);
    }
    auto emit_start = [&](int c) {     // astart of the span that begins at chunk c (if one does)
        if (c % p.cps == 0) {
            const int span = c / p.cps;
            if (span < p.spans) {
#pragma unroll
                for (int j = 0; j < VPL; ++j)
                    p.ework[((size_t)row * p.spans + span) * p.VP + vidx[j]] = asum[j];   // = astart
            }
        }
    };
    for (int c = c_begin; c <= c_end; ++c) {
        if (PARTS == 1) emit_start(c);
        if (c == c_end) break;
        const int n_lo = c * DDSPP_CHUNK, n_hi = min(n_lo + DDSPP_CHUNK, N);
        const int t_lo = n_lo / U, t_hi = min((n_hi - 1) / U + 1, T - 1);
        check_upto(t_hi);
        // no change in (t_lo, t_hi] <=> every frame the chunk touches carries the same frequencies
        const bool chunk_const = (t_checked >= t_hi) && (last_change <= t_lo);
        float e[VPL];
        if (PARTS > 1 && !chunk_const && p.skip_moving) continue;     // bank_scan_kernel writes this chunk's end phases
        if (chunk_const && memo_t >= 0 && last_change <= memo_t) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) e[j] = e_memo[j];
        } else if (chunk_const) {
            float xa[VPL], om[VPL], ph[VPL];
            hf_of(t_lo, xa);
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                om[j] = omega_of<false>(xa[j], p.sr, p.rsr);
                ph[j] = 0.0f;
            }
            for (int n = n_lo; n < n_hi; n += BLK) {          // chunk lengths are multiples of BLK (U and 1000 are)
#pragma unroll
                for (int i = 0; i < BLK; ++i)
#pragma unroll
                    for (int j = 0; j < VPL; ++j) ph[j] = ph[j] + om[j];
            }
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                e[j] = mod_2pi(ph[j]);
                e_memo[j] = e[j];
            }
            memo_t = (n_hi - n_lo == DDSPP_CHUNK) ? t_lo : -1;     // a short last chunk is no memo
        } else {
            // moving frequencies: sample by sample, BLK samples per step (one scalar load of their interpolation
            // weights, requested a step ahead; a step never straddles a frame: U % BLK == 0), the raw controls of the
            // frame after next requested a frame ahead, and the exact constant division when every frequency of the
            // frame pair is in its checked range (same test as the main kernel's classify_frame)
            float ph[VPL], x0[VPL], x1[VPL], qf[VPL], qs[VPL];
#pragma unroll
            for (int j = 0; j < VPL; ++j) ph[j] = 0.0f;
            int tt = t_lo, r = n_lo - t_lo * U;
            hf_of(tt, x0);
            hf_of(min(tt + 1, T - 1), x1);
            hf_raw(min(tt + 2, T - 1), qf, qs);
            auto pair_ok = [&]() {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const float lo = fminf(x0[j], x1[j]), hi = fmaxf(x0[j], x1[j]);
                    ok = ok && (lo > 1e-28f || (lo == 0.0f && (hi == 0.0f || hi > 1e-24f))) && (hi < 3.0e38f);
                }
                return p.fastdiv && __all(ok);
            };
            bool fast = pair_ok();
            const float srv = in_vgpr(p.sr), rsrv = in_vgpr(p.rsr);
            // the interpolation weights go through LDS, PRE_W samples at a time (one coalesced float4 per lane, then two
            // broadcast ds_read_b128 per step, requested a step ahead).  Scalar loads a step ahead were not enough
            // here: a step is 60 instructions, a scalar-cache miss over a thousand cycles.
            auto stage_weights = [&](int n_first) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int idx = lane * 4;
                if (n_first + idx < n_hi)
                    *reinterpret_cast<float4*>(wlds + idx) = *reinterpret_cast<const float4*>(p.wlin + n_first + idx);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            };
            float wl[BLK], wn[BLK];
            auto weights_at = [&](int off, float* w) {
                const float4 wa = *reinterpret_cast<const float4*>(wlds + off);
                const float4 wb = *reinterpret_cast<const float4*>(wlds + off + 4);
                w[0] = wa.x; w[1] = wa.y; w[2] = wa.z; w[3] = wa.w;
                w[4] = wb.x; w[5] = wb.y; w[6] = wb.z; w[7] = wb.w;
            };
            // Frame by frame (round 4): the weights of the frame's samples are staged once (U <= PRE_W), the exact /
            // IEEE division is chosen once per frame pair, and only a frame's LAST block can hold "next row" samples
            // -- the block loop itself has no test but its own (it had five scalar branches per 60-instruction block).
            for (int n = n_lo; n < n_hi;) {
                const int nf = min(n + min(U - r, PRE_W), n_hi);   // end of the frame's part inside the chunk (of PRE_W samples of a longer frame)
                stage_weights(n);                                   // weights of samples n .. n + 255 (the frame's are among them)
                weights_at(0, wl);
                auto blocks = [&](auto fast_tag) {
                    constexpr bool FAST = decltype(fast_tag)::value;
                    int woff = 0;
                    for (; n + BLK < nf; n += BLK) {                // all but the frame's last block
                        weights_at(woff + BLK, wn);
#pragma unroll
                        for (int j = 0; j < VPL; ++j) ph[j] = scan_block_staged<FAST>(ph[j], x0[j], x1[j], wl, srv, rsrv, false);
#pragma unroll
                        for (int i = 0; i < BLK; ++i) wl[i] = wn[i];
                        woff += BLK;
                        r += BLK;
                    }
                    const bool nxt = r + BLK == U &&
                                     __builtin_amdgcn_readfirstlane(__float_as_int(wl[BLK - 1])) == __float_as_int(WALK_NEXT_ROW);
#pragma unroll
                    for (int j = 0; j < VPL; ++j) ph[j] = scan_block_staged<FAST>(ph[j], x0[j], x1[j], wl, srv, rsrv, nxt);
                    n += BLK;
                    r += BLK;
                };
                if (fast) blocks(std::true_type{});
                else blocks(std::false_type{});
                if (r == U) {
                    r = 0;
                    ++tt;
#pragma unroll
                    for (int j = 0; j < VPL; ++j) x0[j] = x1[j];
                    hf_calc(qf, qs, x1);
                    hf_raw(min(tt + 2, T - 1), qf, qs);
                    fast = pair_ok();
                }
            }
#pragma unroll
            for (int j = 0; j < VPL; ++j) e[j] = mod_2pi(ph[j]);
        }
        if (PARTS == 1) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) asum[j] = asum[j] + e[j];
        } else {
#pragma unroll
            for (int j = 0; j < VPL; ++j) p.echunk[((size_t)row * p.npre + c) * p.VP + vidx[j]] = e[j];
        }
    }
}

// launchers of the per-VPL instances: defined here, instantiated by the part that owns the VPL (the others see `extern template`)
template <int VPL>
void launch_prepass_one_wave(const OscParams& q, dim3 grid, dim3 blk, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL((osc_prepass_fused_kernel<VPL, 1>), grid, blk, lds, stream, q);
}
// osc_kernel<VPL, fused, MODE_PREPASS>: the block machinery as a chunk pre-pass (span_starts, whole-file mode with > 128 oscillators per row)
template <int VPL>
void launch_block_prepass(const OscParams& q, dim3 grid, dim3 blk, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL((osc_kernel<VPL, true, MODE_PREPASS, true>), grid, blk, lds, stream, q);
}

#if DDSPP_OSC_PART == 0
// Chunk-parallel pre-pass for a few long rows (a whole file as one segment): one wavefront per (row, chunk) writes
// the chunk's end phase e = (sum of the chunk's 1000 omegas, sequentially in float32) % 2 pi.  All frames the chunk
// touches are fetched in one batch; a chunk whose oscillators keep their frequency (a held note -- nearly all of
// them) is 1000 plain adds of a constant, other chunks interpolate the frequency per sample out of LDS.  Same
// arithmetic as osc_kernel<.., MODE_PREPASS> (which walks every chunk through the full block machinery: 0.50 ms for
// a 136 s file against 0.1 ms here) and as osc_prepass_fused_kernel.
constexpr int PRE_FR = 24;              // frames a chunk may touch (1000 / U + 3)
template <int VPL>
__global__ void __launch_bounds__(256) osc_prepass_chunk_kernel(const OscParams p) {
    extern __shared__ float lds_dyn[];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int task = wave_uniform(blockIdx.x * 4 + wib);
    if (task >= p.R * p.npre) return;
    const int row = task / p.npre, chunk = task - row * p.npre;
    const int T = p.T, U = p.U, H = p.H, S = p.S, N = p.N;
    typedef const __attribute__((address_space(4))) float* cfloat_p;
    const cfloat_p wlin_c = (cfloat_p)(uintptr_t)p.wlin;
    float* hfs = lds_dyn + (size_t)wib * PRE_FR * 64 * VPL;          // [frame][j][lane]
    int vk[VPL], vs[VPL], vidx[VPL];
    bool valid[VPL];
    float kmul[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int v = lane + 64 * j;
        vidx[j] = v;
        valid[j] = v < p.V;
        const int vc = min(v, p.V - 1);
        vs[j] = vc / H;
        vk[j] = vc - vs[j] * H;
        kmul[j] = (float)(vk[j] + 1);
    }
    const int n_lo = chunk * DDSPP_CHUNK, n_hi = min(n_lo + DDSPP_CHUNK, N);
    const int t_lo = n_lo / U;
    const int nfr = (n_hi - 1) / U - t_lo + 2;                       // frames t_lo .. t_last + 1
    // ---- all frames of the chunk in one batch: branch-free addresses, every load issued before the first use
    // shifts from the [R, T, H] tensor, or formed from inharm_coef [R, T]; neither: any finite buffer, ignored
    const bool from_inh = !p.shifts && p.inh;
    const float* shp = p.shifts ? p.shifts : (from_inh ? p.inh : p.hd);
    const size_t sh_stride = from_inh ? 1 : H;
    const float* f0p[VPL];
    const float* shq[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        f0p[j] = p.f0 + (size_t)row * T * S + vs[j];
        shq[j] = shp + (size_t)row * T * sh_stride + (from_inh ? 0 : vk[j]);
    }
    float rf[PRE_FR][VPL], rs[PRE_FR][VPL];
#pragma unroll
    for (int u = 0; u < PRE_FR; ++u) {
        const int fr = wave_uniform(min(t_lo + min(u, nfr - 1), T - 1));
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            rf[u][j] = f0p[j][(size_t)fr * S];
            rs[u][j] = shq[j][(size_t)fr * sh_stride];
        }
    }
    bool same = true;
    float hf0[VPL];
#pragma unroll
    for (int u = 0; u < PRE_FR; ++u)
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            float f = rf[u][j] * kmul[j];
            if (p.shifts) f = f * (1.0f + rs[u][j]);
            else if (from_inh) f = f * (1.0f + shift_from_inharm(rs[u][j], kmul[j]));
            f = valid[j] ? f : 0.0f;
            if (u == 0) hf0[j] = f;
            same = same && (f == hf0[j]);
            hfs[(u * VPL + j) * 64 + lane] = f;
        }
    float ph[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) ph[j] = 0.0f;
    if (__all(same)) {
        float om[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) om[j] = omega_of<false>(hf0[j], p.sr, p.rsr);
        for (int n = n_lo; n < n_hi; n += BLK) {          // chunk lengths are multiples of BLK (U and 1000 are)
#pragma unroll
            for (int i = 0; i < BLK; ++i)
#pragma unroll
                for (int j = 0; j < VPL; ++j) ph[j] = ph[j] + om[j];
        }
    } else {
        float x0[VPL], x1[VPL];
        int tt = 0, r = n_lo - t_lo * U;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            x0[j] = hfs[(0 * VPL + j) * 64 + lane];
            x1[j] = hfs[(1 * VPL + j) * 64 + lane];
        }
        const float srv = in_vgpr(p.sr), rsrv = in_vgpr(p.rsr);
        // BLK samples per step (a step never straddles a frame: U % BLK == 0, chunks start on multiples of BLK); the
        // step's weights are one scalar load, requested a step ahead
        float wl[BLK], wn[BLK];
#pragma unroll
        for (int i = 0; i < BLK; ++i) wl[i] = wlin_c[n_lo + i];
        for (int n = n_lo; n < n_hi; n += BLK) {
            const int nn = min(n + BLK, n_hi - BLK);
#pragma unroll
            for (int i = 0; i < BLK; ++i) wn[i] = wlin_c[nn + i];
            const bool nxt = r + BLK == U && wl[BLK - 1] == WALK_NEXT_ROW;
#pragma unroll
            for (int j = 0; j < VPL; ++j) ph[j] = scan_block_staged<false>(ph[j], x0[j], x1[j], wl, srv, rsrv, nxt);
#pragma unroll
            for (int i = 0; i < BLK; ++i) wl[i] = wn[i];
            r += BLK;
            if (r == U) {
                r = 0;
                ++tt;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    x0[j] = x1[j];
                    x1[j] = hfs[(min(tt + 1, nfr - 1) * VPL + j) * 64 + lane];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VPL; ++j) p.ework[((size_t)row * p.npre + chunk) * p.VP + vidx[j]] = mod_2pi(ph[j]);
}

// rowmax[row] = max over the frames of a row of the per-frame audible-harmonic counts: what the pre-pass and the scans
// behind it ask before they touch a 64-oscillator group ("is any of it ever heard?").  One workgroup per row: asked by
// every wavefront of a 136 s row (34 000 frames) it was 45 memory round trips each.
__global__ void __launch_bounds__(256) osc_row_max_kernel(const int* __restrict__ audible, int* __restrict__ rowmax,
                                                        int T) {
    __shared__ int part[4];
    const int row = blockIdx.x, lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int* aud = audible + (size_t)row * T;
    int amax = 0;
    for (int t0 = wib * 64 + lane; t0 < T; t0 += 256 * 16) {
        int a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] = aud[min(t0 + 256 * u, T - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) amax = max(amax, a[u] & 0xffff);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = max(amax, __shfl_xor(amax, o));
    if (lane == 0) part[wib] = amax;
    __syncthreads();
    if (threadIdx.x == 0) rowmax[row] = max(max(part[0], part[1]), max(part[2], part[3]));
}

// Sequential (float32) scan of the chunk end phases: astart[row, span, v] = e[0] + ... + e[c0-1]
// in exactly the order of `tf.cumsum(offsets, axis=1)` in ddsp.core.angular_cumsum -- for LONG rows (a whole file as one
// segment: thousands of chunks).  The adds of one (row, oscillator) are a serial chain, the loads and stores are not: a
// workgroup owns 64 oscillators of a row, its four wavefronts fetch the next SCAN_CT chunks x 64 values into registers
// (all loads in flight) while wavefront 0 turns the current tile in LDS into running sums IN PLACE, and all four then
// store the span starts of the tile.  (A thread per chain with its own loads took 0.38 ms for a 136 s file: sixteen rows
// are twelve wavefronts, each paying the memory latency two hundred times; with wavefront 0 also issuing the 3 263 stores
// of its chain one after the other, 0.24 ms.)  Groups no partial of which is audible anywhere in the call are left out
// (p.audible / p.rowmax given, need_all not set): nothing reads their start phases.
constexpr int SCAN_CT = 256;           // chunks per tile: 64 KB of LDS, 64 registers per thread
__global__ void __launch_bounds__(256) osc_offset_scan_kernel(const float* __restrict__ ework,
                                                            float* __restrict__ astart, const OscParams p) {
    __shared__ float tile[SCAN_CT * 64];
    const int R = p.R, npre = p.npre, VP = p.VP, spans = p.spans, cps = p.cps;
    (void)R;
    const int groups = VP / 64;
    const int row = blockIdx.x / groups, v0 = (blockIdx.x - row * groups) * 64;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    if (p.audible && !p.need_all) {
        const int amax = p.rowmax ? p.rowmax[row] : row_audible_max(p.audible + (size_t)row * p.T, p.T, lane);
        const int v = v0 + lane;
        if (!__any(v < p.V && v % p.H < amax)) return;        // (the same for every wavefront of the workgroup)
    }
    constexpr int PER = SCAN_CT / 4;                      // chunks each wavefront fetches per tile
    float pre[PER];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = min(c0 + wib * PER + u, npre - 1);
            pre[u] = ework[((size_t)row * npre + c) * VP + v0 + lane];
        }
    };
    fetch(0);
    float a = (p.state_in && v0 + lane < p.V) ? p.state_in[(size_t)row * p.V + v0 + lane] : 0.0f;
    float* dst = astart + (size_t)row * spans * VP + v0 + lane;
    for (int c0 = 0; c0 <= npre; c0 += SCAN_CT) {
#pragma unroll
        for (int u = 0; u < PER; ++u) tile[(wib * PER + u) * 64 + lane] = pre[u];
        __syncthreads();
        if (c0 + SCAN_CT <= npre) fetch(c0 + SCAN_CT);     // in flight while wavefront 0 scans
        if (wib == 0) {
            if (c0 + SCAN_CT <= npre) {                    // a full tile: no bound to test per chunk
                for (int u0 = 0; u0 < SCAN_CT; u0 += 16) {
                    float e[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) e[u] = tile[(u0 + u) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        tile[(u0 + u) * 64 + lane] = a;                    // the sum BEFORE chunk c: where a span at c starts
                        a = a + e[u];
                    }
                }
            } else {
                for (int u0 = 0; u0 < SCAN_CT; u0 += 16) {
                    if (c0 + u0 > npre) break;
                    float e[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) e[u] = tile[(u0 + u) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        tile[(u0 + u) * 64 + lane] = a;
                        if (c0 + u0 + u < npre) a = a + e[u];
                    }
                }
            }
        }
        __syncthreads();
        // span starts of this tile, stored by all four wavefronts: span s starts at chunk s * cps
        {
            const int c_last = min(c0 + SCAN_CT - 1, npre);
            const int s_end = min(spans, c_last / cps + 1);                // spans that start in this tile: [s_first, s_end)
            int s_ = (c0 + cps - 1) / cps + wib;
            for (; s_ + 28 < s_end; s_ += 32) {                            // eight LDS reads in flight, then eight stores
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = tile[((s_ + 4 * k) * cps - c0) * 64 + lane];
#pragma unroll
                for (int k = 0; k < 8; ++k) dst[(size_t)(s_ + 4 * k) * VP] = v[k];
            }
            for (; s_ < s_end; s_ += 4) dst[(size_t)s_ * VP] = tile[(s_ * cps - c0) * 64 + lane];
        }
        __syncthreads();
    }
}

// Few chunks (a 3 s segment has 72): one thread per chain, loads batched by 16.
__global__ void __launch_bounds__(256) osc_offset_scan_short_kernel(const float* __restrict__ ework,
                                                                  float* __restrict__ astart, int R,
                                                                  int npre, int VP, int spans, int cps,
                                                                  const float* __restrict__ state_in, int V) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)R * VP) return;
    const int row = (int)(gid / VP), v = (int)(gid - (size_t)row * VP);
    float a = (state_in && v < V) ? state_in[(size_t)row * V + v] : 0.0f;
    int span = 0;
    constexpr int NB = 16;
    for (int c0 = 0; c0 <= npre; c0 += NB) {
        float e[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) e[u] = ework[((size_t)row * npre + min(c0 + u, max(npre - 1, 0))) * VP + v];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int c = c0 + u;
            if (c <= npre && c == span * cps && span < spans) {
                astart[((size_t)row * spans + span) * VP + v] = a;
                ++span;
            }
            if (c < npre) a = a + e[u];
        }
    }
}

// The scan behind the sectioned memo pre-pass (osc_prepass_fused_kernel<.., 4>): one wavefront per (row, 64 oscillators),
// the sixty-fours the pre-pass left out (no partial of theirs audible anywhere in the call, same test) left out here
// too -- nothing reads their start phases.
__global__ void __launch_bounds__(256) osc_offset_scan_groups_kernel(const float* __restrict__ echunk,
                                                                   float* __restrict__ astart, OscParams p) {
    const int lane = threadIdx.x & 63;
    const int task = wave_uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int cols = p.VP / 64;
    if (task >= p.R * cols) return;
    const int row = task / cols;
    const int v = (task - row * cols) * 64 + lane;
    if (p.audible && !p.need_all) {
        const int amax = p.rowmax ? p.rowmax[row] : row_audible_max(p.audible + (size_t)row * p.T, p.T, lane);
        if (!__any(v < p.V && v % p.H < amax)) return;
    }
    constexpr int NB = 40;
    const int npre = p.npre;
    const float* src = echunk + (size_t)row * npre * p.VP + v;
    float e[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) e[u] = src[(size_t)min(u, max(npre - 1, 0)) * p.VP];
    float a = (p.state_in && v < p.V) ? p.state_in[(size_t)row * p.V + v] : 0.0f;
    int span = 0;
    for (int c0 = 0; c0 <= npre; c0 += NB) {
        float en[NB];
        if (c0 + NB <= npre) {
#pragma unroll
            for (int u = 0; u < NB; ++u) en[u] = src[(size_t)min(c0 + NB + u, npre - 1) * p.VP];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int c = c0 + u;
            if (c <= npre && c == span * p.cps && span < p.spans) {
                astart[((size_t)row * p.spans + span) * p.VP + v] = a;
                ++span;
            }
            if (c < npre) a = a + e[u];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) e[u] = en[u];
    }
}

void launch_offset_scan(const float* ework, float* astart, int R, int npre, int VP, int spans, int cps,
                        hipStream_t stream, const float* state_in, int V) {
    const size_t nthr = (size_t)R * VP;
    if (npre > SCAN_CT && !env_int("DDSPP_OSC_SHORT_SCAN", 0)) {
        OscParams q{};                     // (no audible counts on this path: every group is scanned)
        q.R = R; q.npre = npre; q.VP = VP; q.spans = spans; q.cps = cps; q.state_in = state_in; q.V = V; q.H = V > 0 ? V : 1;
        hipLaunchKernelGGL(osc_offset_scan_kernel, dim3((unsigned)(nthr / 64)), dim3(256), 0, stream, ework, astart, q);
    } else {
        hipLaunchKernelGGL(osc_offset_scan_short_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, stream,
                           ework, astart, R, npre, VP, spans, cps, state_in, V);
    }
}

// nk[b, span, p] = number of leading harmonics of voice p that have a non-zero amplitude
// amp[t] * hd[t, k] in some frame the span touches (frames t_lo .. min(t_hi + 1, T - 1)).
// get_controls zeroes harmonics above Nyquist and gates silent voices, so for a piano note this is
// floor(Nyquist / f_k) -- typically a third of H.
__global__ void __launch_bounds__(256) osc_count_kernel(const float* __restrict__ amp, const float* __restrict__ hd,
                                                      int* __restrict__ nk, int R, int P, int T, int H, int U,
                                                      int N, int spans, int cps, int vmajor) {
    const int lane = threadIdx.x & 63;
    const int task = wave_uniform(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (task >= R * spans) return;
    const int row = task / spans, span = task - row * spans;
    const int n_lo = span * cps * DDSPP_CHUNK, n_hi = min((span + 1) * cps * DDSPP_CHUNK, N);
    const int t_lo = n_lo / U, t_hi = min((n_hi - 1) / U + 1, T - 1);
    int best = 0;
    constexpr int FB = 8;
    for (int k0 = 0; k0 < H; k0 += 64) {
        const int k = min(k0 + lane, H - 1);
        bool any = false;
        for (int t0 = t_lo; t0 <= t_hi; t0 += FB) {
            float a[FB], h[FB];
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const size_t fr = (size_t)row * T + min(t0 + u, t_hi);
                a[u] = amp[fr];
                h[u] = hd[fr * H + k];
            }
#pragma unroll
            for (int u = 0; u < FB; ++u) any = any || (a[u] * h[u] != 0.0f);
        }
        if (any && k0 + lane < H) best = k0 + lane + 1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    if (lane == 0) {
        const int B = R / P;
        const int b = vmajor ? row % B : row / P, v = vmajor ? row / B : row - b * P;
        nk[((size_t)b * spans + span) * P + v] = best;
    }
}

// The same from the per-frame counts ddspp_inharmonic_controls leaves behind (audible[R, T]): one thread per
// (row, span), a dozen ints instead of a [frames, H] scan.
__global__ void __launch_bounds__(256) osc_count_frames_kernel(const int* __restrict__ audible, int* __restrict__ nk,
                                                             int R, int P, int T, int U, int N, int spans, int cps,
                                                             int vmajor) {
    const int task = blockIdx.x * 256 + threadIdx.x;
    if (task >= R * spans) return;
    const int row = task / spans, span = task - row * spans;
    const int n_lo = span * cps * DDSPP_CHUNK, n_hi = min((span + 1) * cps * DDSPP_CHUNK, N);
    const int t_lo = n_lo / U, t_hi = min((n_hi - 1) / U + 1, T - 1);
    int best = 0;
    for (int t = t_lo; t <= t_hi; ++t) best = max(best, audible[(size_t)row * T + t] & 0xffff);
    const int B = R / P;
    const int b = vmajor ? row % B : row / P, v = vmajor ? row / B : row - b * P;
    nk[((size_t)b * spans + span) * P + v] = best;
}

// The same with the compacted scan of moving chunks in view (round 5): ONE WAVEFRONT PER ROW, lane = span (+ 64, + 128, ...),
// so that what the scan needs comes out of wave reductions with no atomics and no buffer to zero first:
//   rowmax[row]        the row's audible maximum over all its spans (what the pre-pass and the scans ask of a 64-group)
//   moved[row, c]      1 when the row's frequencies move in chunk c < npre (bit 16 of a frame's count: its frequencies differ
//                      from the frame before; the test is osc_prepass_fused_kernel's `chunk_const`: no change in frames
//                      (t_lo, t_hi] of the chunk) -- written for every (row, chunk), 0 or 1
//   *any = call_id     when some row moves somewhere: bank_scan_kernel leaves at once unless it finds this call's id there
//                      (a stale or garbage value that happens to match only costs a walk over flags that are all current)
__global__ void __launch_bounds__(64) osc_count_rows_kernel(const int* __restrict__ audible, int* __restrict__ nk, int R, int P,
                                                          int T, int U, int N, int spans, int cps, int vmajor,
                                                          int* __restrict__ rowmax_out, int* __restrict__ moved_out,
                                                          int* __restrict__ any, int call_id, int npre) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const int B = R / P;
    const int b = vmajor ? row % B : row / P, v = vmajor ? row / B : row - b * P;
    int rmax = 0, any_moved = 0;
    for (int span = lane; span < spans; span += 64) {
        const int n_lo = span * cps * DDSPP_CHUNK, n_hi = min((span + 1) * cps * DDSPP_CHUNK, N);
        const int t_lo = n_lo / U, t_hi = min((n_hi - 1) / U + 1, T - 1);
        int best = 0;
        for (int t = t_lo; t <= t_hi; ++t) best = max(best, audible[(size_t)row * T + t] & 0xffff);
        nk[((size_t)b * spans + span) * P + v] = best;
        rmax = max(rmax, best);
        for (int c = span * cps; c < min((span + 1) * cps, npre); ++c) {
            const int cn_lo = c * DDSPP_CHUNK, cn_hi = min(cn_lo + DDSPP_CHUNK, N);
            const int ct_lo = cn_lo / U, ct_hi = min((cn_hi - 1) / U + 1, T - 1);
            int moved = 0;
            for (int t = ct_lo + 1; t <= ct_hi; ++t) moved |= audible[(size_t)row * T + t] >> 16;
            moved_out[(size_t)row * npre + c] = moved & 1;
            any_moved |= moved & 1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rmax = max(rmax, __shfl_xor(rmax, o));
    if (lane == 0) rowmax_out[row] = rmax;
    if (__any(any_moved) && lane == 0) *any = call_id;
}

static bool sample_rate_is_checked(float sr) {
    // rates for which div_const == IEEE division was verified for every float32 input with
    // |x| >= 1e-28 (tests/test_exact_arith.py builds and runs the checker)
    const float ok[] = {8000.f, 16000.f, 22050.f, 24000.f, 32000.f, 44100.f, 48000.f, 96000.f};
    for (float v : ok)
        if (sr == v) return true;
    return false;
}

int env_int(const char* name, int dflt) { return ddspp_option_literal(name, dflt); }

static int pick_vpl(int V) {
    const int need = (V + 63) / 64;
    const int avail[] = {1, 2, 3, 4, 6, 8};
    for (int a : avail)
        if (a >= need) return a;
    return 0;
}

// materialised source: prefer a VPL that divides H (vector loads, lane owns VPL adjacent sinusoids)
static int pick_vpl_materialised(int H) {
    const int need = (H + 63) / 64;
    const int contig[] = {1, 2, 4};
    for (int a : contig)
        if (a >= need && H % a == 0) return a;
    const int strided[] = {3, 6, 8};
    for (int a : strided)
        if (a >= need) return a;
    return 0;
}

struct Plan {
    int vpl, VP, nchunks, spans, cps, npre, groups, vgrp;
    size_t ework_floats, astart_floats, partial_floats;
};

static Plan make_plan(int R, int N, int V, bool angular, bool fused, int spans_req, int groups_req,
                      bool sum) {
    Plan pl{};
    // Split a row's oscillators over several wavefronts when the rows alone leave SIMDs with a
    // single wavefront (nothing to overlap its memory waits with).  Groups are multiples of 64.
    int groups = 1;
    const int max_groups = min((V + 63) / 64, 4);          // one workgroup (<= 256 threads) per row
    if (groups_req > 0) groups = groups_req;
    else if (sum && !fused) {
        // HBM-bound source: two wavefronts per SIMD overlap one's vmcnt waits with the other's ALU work
        const int want = env_int("DDSPP_OSC_GROUP_WAVES", 2048);
        groups = (want + R - 1) / R;
    } else if (sum) {
        // ALU-bound source: spans give finer, better balanced tasks; split rows only for tiny batches
        const long long tasks = (long long)R * ((N + DDSPP_CHUNK - 1) / DDSPP_CHUNK);
        const int want = env_int("DDSPP_OSC_GROUP_WAVES_FUSED", 4096);
        groups = tasks >= want ? 1 : (int)((want + tasks - 1) / tasks);
    }
    if (groups > max_groups) groups = max_groups;
    if (groups < 1 || !sum) groups = 1;
    pl.vgrp = ((V + groups - 1) / groups + 63) / 64 * 64;
    if (groups == 1) pl.vgrp = V;
    pl.groups = (V + pl.vgrp - 1) / pl.vgrp;
    pl.vpl = fused ? pick_vpl(pl.vgrp) : pick_vpl_materialised(pl.vgrp);
    pl.VP = pl.groups * pl.vgrp > pl.vpl * 64 ? (pl.groups * pl.vgrp + 63) / 64 * 64 : pl.vpl * 64;
    R = R * pl.groups;
    pl.nchunks = (N + DDSPP_CHUNK - 1) / DDSPP_CHUNK;
    int spans = 1;
    if (angular) {
        if (spans_req > 0) {
            spans = spans_req;
        } else {
            // enough wavefronts to occupy 256 CUs; the fused source is ALU bound and likes more
            // waves, the materialised source pays one extra read of `fe` per pre-passed chunk and
            // is kept at one span whenever the rows alone give >= 4 waves per CU.
            const int target = fused ? env_int("DDSPP_OSC_TARGET_WAVES_FUSED", 36864)
                                     : env_int("DDSPP_OSC_TARGET_WAVES", 1024);
            spans = (target + R - 1) / R;
        }
        if (spans < 1) spans = 1;
        if (spans > pl.nchunks) spans = pl.nchunks;
    }
    pl.cps = (pl.nchunks + spans - 1) / spans;
    pl.spans = (pl.nchunks + pl.cps - 1) / pl.cps;
    pl.npre = pl.spans > 1 ? (pl.spans - 1) * pl.cps : 0;
    const int rows = R / pl.groups;
    pl.ework_floats = (size_t)rows * pl.npre * pl.VP;
    pl.astart_floats = (size_t)rows * pl.spans * pl.VP;
    pl.partial_floats = 0;
    return pl;
}

// Launch of the memoised pre-pass: `tasks` (row, group) pairs, start offsets into q.ework (= astart).  In sections
// (workgroups of four wavefronts, PARTS = 4, chunk end phases through `echunk` [R, npre, VP] and the group scan) when the
// task is at most 128 oscillators wide and there are chunks enough to share out; one wavefront per pair otherwise.
void launch_memo_prepass(int vpl, const OscParams& q0, int tasks, float* echunk, hipStream_t stream) {
    const size_t ldsw = (size_t)4 * PRE_W * sizeof(float);              // weight buffers of the four wavefronts
    const bool parts4 = vpl <= 2 && echunk && q0.npre >= 8 && !env_int("DDSPP_OSC_PREPASS_ONE_WAVE", 0);
    // (skip_moving is only set where this holds: polyphonic_additive_impl tests the same conditions)
    if (parts4) {
        OscParams q = q0;
        // runs of about six chunks per wavefront: short enough to balance, long enough that the held-note case (one
        // memoised scan of 1000 dependent adds and four memory latencies per wavefront, whatever its run) stays cheap:
        // 3 s rows, runs of 3 / 6 / 9 chunks: held notes 62 / 41 / 35 us, every frame moving 0.90 / 0.94 / 0.98 ms
        const int per_wave = max(env_int("DDSPP_OSC_PREPASS_RUN", 6), 1);
        q.nsec = max((q.npre + 4 * per_wave - 1) / (4 * per_wave), 1);
        q.echunk = echunk;
        if (vpl == 1) hipLaunchKernelGGL((osc_prepass_fused_kernel<1, 4>), dim3(tasks * q.nsec), dim3(256), ldsw, stream, q);
        else hipLaunchKernelGGL((osc_prepass_fused_kernel<2, 4>), dim3(tasks * q.nsec), dim3(256), ldsw, stream, q);
        if (q.skip_moving) {               // the chunks the pre-pass left alone (frequencies moving), packed like the bank
            OscParams b = q;
            b.R = q.R / q.P;               // segments
            launch_bank_scan(b, env_int("DDSPP_OSC_SCAN_VPL", 2), stream);
        }
        if (q.npre > SCAN_CT)              // long rows: tiles through LDS, four wavefronts per (row, 64 oscillators)
            hipLaunchKernelGGL(osc_offset_scan_kernel, dim3(q.R * (q.VP / 64)), dim3(256), 0, stream, echunk, q.ework, q);
        else
            hipLaunchKernelGGL(osc_offset_scan_groups_kernel, dim3((q.R * (q.VP / 64) + 3) / 4), dim3(256), 0, stream, echunk,
                               q.ework, q);
        return;
    }
    const OscParams& q = q0;
    const dim3 grid((tasks + 3) / 4), blk(256);
    switch (vpl) {
        case 1: launch_prepass_one_wave<1>(q, grid, blk, ldsw, stream); break;
        case 2: launch_prepass_one_wave<2>(q, grid, blk, ldsw, stream); break;
        case 3: launch_prepass_one_wave<3>(q, grid, blk, ldsw, stream); break;
        case 4: launch_prepass_one_wave<4>(q, grid, blk, ldsw, stream); break;
        case 6: launch_prepass_one_wave<6>(q, grid, blk, ldsw, stream); break;
        default: launch_prepass_one_wave<8>(q, grid, blk, ldsw, stream); break;
    }
}
#endif  // DDSPP_OSC_PART == 0

template <int VPL, bool FUSED, bool DECAY = false>
void launch_all(const OscParams& p, bool angular, bool sum, hipStream_t stream) {
    const int nblk_main = p.R * p.spans;
    const dim3 blk(64 * p.groups);
    const size_t lds = ((size_t)p.groups * (TILE * TSTRIDE) + 2 * p.groups * 32) * sizeof(float);
    if (angular) {
        // The memoised pre-pass walks the chunks of a row sequentially: right when there are enough
        // rows to fill the chip, wrong for a single long file (few rows, thousands of chunks), where the
        // chunk-parallel pre-pass + offset scan is used instead.
        const bool memo = FUSED && p.R * p.groups >= env_int("DDSPP_OSC_MEMO_MIN_WAVES", 256) &&
                          !env_int("DDSPP_OSC_PLAIN_PREPASS", 0);
        if (p.spans > 1 && memo) {
            OscParams q = p;
            q.ework = const_cast<float*>(p.astart);      // the memo pre-pass writes astart directly
            launch_memo_prepass(VPL, q, p.R * p.groups, p.ework, stream);
        } else if (p.spans > 1) {
            const int nblk_pre = p.R * p.npre;
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_PREPASS, true>), dim3(nblk_pre), blk, lds,
                               stream, p);
            launch_offset_scan(p.ework, const_cast<float*>(p.astart), p.R, p.npre, p.VP, p.spans, p.cps, stream);
        }
        if constexpr (DECAY)               // (phases do not depend on the decay term: the pre-pass is the plain one)
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_MAIN, true, true>), dim3(nblk_main), blk, lds, stream, p);
        else if (!FUSED && sum && osc_stream_applies(p) && env_int("DDSPP_OSC_STREAM", 1))
            launch_osc_stream(p, stream);          // osc_stream.hip: the same sums, written for the HBM-bound shape
        else if (sum)
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_MAIN, true>), dim3(nblk_main), blk, lds, stream, p);
        else
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_MAIN, false>), dim3(nblk_main), blk, lds, stream,
                               p);
    } else {
        if constexpr (DECAY)
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_PLAIN, true, true>), dim3(nblk_main), blk, lds, stream, p);
        else if (sum)
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_PLAIN, true>), dim3(nblk_main), blk, lds, stream,
                               p);
        else
            hipLaunchKernelGGL((osc_kernel<VPL, FUSED, MODE_PLAIN, false>), dim3(nblk_main), blk, lds, stream,
                               p);
    }
}

#define DDSPP_OSC_INSTANCES(vpl, linkage)                                                                          \
    linkage template void launch_all<vpl, false, false>(const OscParams&, bool, bool, hipStream_t);                    \
    linkage template void launch_all<vpl, true, false>(const OscParams&, bool, bool, hipStream_t);                     \
    linkage template void launch_all<vpl, true, true>(const OscParams&, bool, bool, hipStream_t);                      \
    linkage template void launch_prepass_one_wave<vpl>(const OscParams&, dim3, dim3, size_t, hipStream_t);             \
    linkage template void launch_block_prepass<vpl>(const OscParams&, dim3, dim3, size_t, hipStream_t);
#if DDSPP_OSC_OWNS(1)
DDSPP_OSC_INSTANCES(1, )
DDSPP_OSC_INSTANCES(2, )
#endif
#if DDSPP_OSC_OWNS(3)
DDSPP_OSC_INSTANCES(3, )
DDSPP_OSC_INSTANCES(4, )
#else
DDSPP_OSC_INSTANCES(3, extern)
DDSPP_OSC_INSTANCES(4, extern)
#endif
#if DDSPP_OSC_OWNS(6)
DDSPP_OSC_INSTANCES(6, )
#else
DDSPP_OSC_INSTANCES(6, extern)
#endif
#if DDSPP_OSC_OWNS(8)
DDSPP_OSC_INSTANCES(8, )
#else
DDSPP_OSC_INSTANCES(8, extern)
#endif

#if DDSPP_OSC_PART == 0
template <bool FUSED, bool DECAY = false>
static int dispatch_vpl(int vpl, const OscParams& p, bool angular, bool sum, hipStream_t stream) {
    switch (vpl) {
        case 1: launch_all<1, FUSED, DECAY>(p, angular, sum, stream); break;
        case 2: launch_all<2, FUSED, DECAY>(p, angular, sum, stream); break;
        case 3: launch_all<3, FUSED, DECAY>(p, angular, sum, stream); break;
        case 4: launch_all<4, FUSED, DECAY>(p, angular, sum, stream); break;
        case 6: launch_all<6, FUSED, DECAY>(p, angular, sum, stream); break;
        case 8: launch_all<8, FUSED, DECAY>(p, angular, sum, stream); break;
        default: return DDSPP_EINVAL;
    }
    return DDSPP_OK;
}

// Span start offsets astart[row, span, v] for every (row, oscillator) of R rows (p carries the plan: spans, cps, npre,
// and the optional streaming state the sums start from): memoised sequential walk when the rows fill the chip,
// chunk-parallel pre-pass + scan for a few long rows (whole-file mode).
static void span_starts(const OscParams& p, int R, int V, int vpl_pre, float* astart, float* ework, hipStream_t stream) {
    const int VP = p.VP, U = p.U;
    OscParams q = p;
    q.R = R; q.groups = 1; q.vgrp = V;
    // The memoised walk needs wavefronts enough to fill the chip: rows did that alone (batches), since the sections of
    // round 3 a few LONG rows do too (a 136 s file: 16 rows x 2 groups x 136 sections)
    const int per_wave = max(env_int("DDSPP_OSC_PREPASS_RUN", 6), 1);
    const long long sections = V % 64 == 0 && pick_vpl(64) == 1 && q.npre >= 8
                                   ? (long long)R * (V / 64) * ((q.npre + 4 * per_wave - 1) / (4 * per_wave)) : 0;
    const bool memo = (R >= env_int("DDSPP_OSC_MEMO_MIN_WAVES", 256) || sections >= env_int("DDSPP_OSC_MEMO_MIN_WAVES", 256)) &&
                      !env_int("DDSPP_OSC_PLAIN_PREPASS", 0);
    if (memo) {
        q.ework = astart;
        // one wavefront per 64 oscillators of a row when the rows alone leave SIMDs with a single wavefront: the walk
        // is a chain of memory latencies (frames in batches), and 64 loads per batch fit the 63-deep load counter
        const bool split = V % 64 == 0 && (long long)R * (V / 64) <= 8192 && !env_int("DDSPP_OSC_PREPASS_WHOLE_ROWS", 0);
        const int tasks = split ? R * (V / 64) : R;
        if (split) {
            q.groups = V / 64;
            q.vgrp = 64;
        }
        launch_memo_prepass(split ? 1 : vpl_pre, q, tasks, ework, stream);
        return;
    }
    q.ework = ework;
    if (q.npre > 0) {
        const bool chunk_kernel = vpl_pre <= 2 && DDSPP_CHUNK / U + 3 <= PRE_FR && !env_int("DDSPP_OSC_OLD_CHUNK_PREPASS", 0);
        if (chunk_kernel) {
            const unsigned wgs = (unsigned)((R * q.npre + 3) / 4);
            const size_t clds = (size_t)4 * PRE_FR * 64 * vpl_pre * sizeof(float);
            if (vpl_pre == 1) hipLaunchKernelGGL((osc_prepass_chunk_kernel<1>), dim3(wgs), dim3(256), clds, stream, q);
            else hipLaunchKernelGGL((osc_prepass_chunk_kernel<2>), dim3(wgs), dim3(256), clds, stream, q);
        } else {
            const dim3 grid((unsigned)(R * q.npre)), blk(64);
            const size_t plds = ((size_t)(TILE * TSTRIDE) + 2 * 32) * sizeof(float);
            switch (vpl_pre) {
                case 1: launch_block_prepass<1>(q, grid, blk, plds, stream); break;
                case 2: launch_block_prepass<2>(q, grid, blk, plds, stream); break;
                case 3: launch_block_prepass<3>(q, grid, blk, plds, stream); break;
                case 4: launch_block_prepass<4>(q, grid, blk, plds, stream); break;
                case 6: launch_block_prepass<6>(q, grid, blk, plds, stream); break;
                default: launch_block_prepass<8>(q, grid, blk, plds, stream); break;
            }
        }
    }
    launch_offset_scan(ework, astart, R, q.npre, VP, p.spans, p.cps, stream, p.state_in, V);
}

#endif  // DDSPP_OSC_PART == 0 (dispatch, span starts)

}  // namespace ddspp

#if DDSPP_OSC_PART == 0
using namespace ddspp;

extern "C" {

// Workspace (bytes) the two oscillator entry points may need for R rows of N samples and V
// oscillators per row (V = n_substrings * n_harmonics for the fused entry point).
size_t ddspp_osc_workspace_bytes(int R, int N, int V) {
    if (R <= 0 || N <= 0 || V <= 0) return 0;
    const int vpl = pick_vpl(V) > pick_vpl_materialised(V) ? pick_vpl(V) : pick_vpl_materialised(V);
    if (!vpl) return 0;
    const size_t nchunks = (N + DDSPP_CHUNK - 1) / DDSPP_CHUNK;
    const size_t vp = ((size_t)V + 127) / 64 * 64 + (size_t)vpl * 64;
    return 2 * (size_t)R * nchunks * vp * sizeof(float) + 1024;
}

// cos_oscillator_bank(frequency_envelopes, amplitude_envelopes, sample_rate, sum_sinusoids,
// use_angular_cumsum)   -- ddsp_piano/modules/inharm_synth.py:49-84
int ddspp_cos_oscillator_bank(const float* frequency_envelopes, const float* amplitude_envelopes,
                              float* audio, int R, int N, int H, float sample_rate, int sum_sinusoids,
                              int use_angular_cumsum, int spans, void* workspace,
                              size_t workspace_bytes, hipStream_t stream) {
    DDSPP_REQUIRE(frequency_envelopes && amplitude_envelopes && audio, "cos_oscillator_bank: null buffer");
    DDSPP_REQUIRE(R > 0 && N > 0 && H > 0, "cos_oscillator_bank: bad dims R=%d N=%d H=%d", R, N, H);
    DDSPP_REQUIRE(pick_vpl_materialised(H) != 0, "cos_oscillator_bank: n_sinusoids=%d exceeds 512", H);
    DDSPP_REQUIRE(N % BLK == 0, "cos_oscillator_bank: n_samples=%d must be a multiple of %d", N, BLK);
    DDSPP_REQUIRE(sample_rate > 0.f, "cos_oscillator_bank: bad sample_rate");
    Plan pl = make_plan(R, N, H, use_angular_cumsum != 0, false,
                        spans > 0 ? spans : env_int("DDSPP_OSC_SPANS", 0), env_int("DDSPP_OSC_GROUPS", 0),
                        sum_sinusoids != 0);
    const size_t need = (pl.ework_floats + pl.astart_floats + pl.partial_floats) * sizeof(float);
    DDSPP_REQUIRE(pl.spans == 1 || (workspace && workspace_bytes >= need),
                  "cos_oscillator_bank: workspace too small (%zu < %zu)", workspace_bytes, need);
    OscParams p{};
    p.fe = frequency_envelopes;
    p.ae = amplitude_envelopes;
    p.out = audio;
    p.ework = (float*)workspace;
    p.astart = p.ework ? p.ework + pl.ework_floats : nullptr;
    p.partial = p.ework ? p.ework + pl.ework_floats + pl.astart_floats : nullptr;
    p.groups = pl.groups; p.vgrp = pl.vgrp;
    p.R = R; p.N = N; p.T = 0; p.U = BLK; p.H = H; p.S = 1; p.V = H; p.VP = pl.VP;
    p.spans = pl.spans; p.cps = pl.cps; p.nchunks = pl.nchunks; p.npre = pl.npre;
    p.sr = sample_rate; p.rsr = 1.0f / sample_rate; p.nyq = sample_rate / 2.0f;
    p.fastdiv = sample_rate_is_checked(sample_rate) && !env_int("DDSPP_NO_FASTDIV", 0);
    p.off_plain = env_int("DDSPP_ANGULAR_OFFSETS_PLAIN", 0) ? 1 : 0;
    int rc = dispatch_vpl<false>(pl.vpl, p, use_angular_cumsum != 0, sum_sinusoids != 0, stream);
    DDSPP_REQUIRE(rc == DDSPP_OK, "cos_oscillator_bank: dispatch failed");
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// harmonic_synthesis(...) summed over the substrings of MultiInharmonic.get_signal, straight from
// the frame-rate controls -- inharm_synth.py:87-127, :272-293.  `wlin` and `whann` come from
// ddspp_resample_tables (resample.hip); the [R, N, H] envelopes are never materialised.
static int harmonic_synthesis_impl(const float* f0_hz, const float* amplitudes,
                                   const float* harmonic_distribution, const float* harmonic_shifts,
                                   const float* decays, const float* decay_time,
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
                        spans > 0 ? spans : env_int("DDSPP_OSC_SPANS_FUSED", 0),
                        env_int("DDSPP_OSC_GROUPS_FUSED", 0), true);
    const size_t need = (pl.ework_floats + pl.astart_floats + pl.partial_floats) * sizeof(float);
    DDSPP_REQUIRE(pl.spans == 1 || (workspace && workspace_bytes >= need),
                  "harmonic_synthesis: workspace too small (%zu < %zu)", workspace_bytes, need);
    OscParams p{};
    p.f0 = f0_hz; p.amp = amplitudes; p.hd = harmonic_distribution; p.shifts = harmonic_shifts;
    p.wlin = wlin; p.whann = whann;
    p.out = audio;
    p.ework = (float*)workspace;
    p.astart = p.ework ? p.ework + pl.ework_floats : nullptr;
    p.partial = p.ework ? p.ework + pl.ework_floats + pl.astart_floats : nullptr;
    p.groups = pl.groups; p.vgrp = pl.vgrp;
    p.R = R; p.N = N; p.T = T; p.U = U; p.H = H; p.S = S; p.V = V; p.VP = pl.VP;
    p.spans = pl.spans; p.cps = pl.cps; p.nchunks = pl.nchunks; p.npre = pl.npre;
    p.sr = sample_rate; p.rsr = 1.0f / sample_rate; p.nyq = sample_rate / 2.0f;
    p.fastdiv = sample_rate_is_checked(sample_rate) && !env_int("DDSPP_NO_FASTDIV", 0);
    p.off_plain = env_int("DDSPP_ANGULAR_OFFSETS_PLAIN", 0) ? 1 : 0;
    p.decays = decays; p.decay_time = decay_time;
    int rc = decays ? dispatch_vpl<true, true>(pl.vpl, p, use_angular_cumsum != 0, true, stream)
                    : dispatch_vpl<true>(pl.vpl, p, use_angular_cumsum != 0, true, stream);
    DDSPP_REQUIRE(rc == DDSPP_OK, "harmonic_synthesis: dispatch failed");
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

int ddspp_harmonic_synthesis(const float* f0_hz, const float* amplitudes,
                             const float* harmonic_distribution, const float* harmonic_shifts,
                             const float* wlin, const float* whann, float* audio, int R, int T, int S,
                             int H, int U, float sample_rate, int use_angular_cumsum, int spans,
                             void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return harmonic_synthesis_impl(f0_hz, amplitudes, harmonic_distribution, harmonic_shifts, nullptr, nullptr, wlin, whann,
                                   audio, R, T, S, H, U, sample_rate, use_angular_cumsum, spans, workspace, workspace_bytes,
                                   stream);
}

// surrogate_harmonic_synthesis -- ddsp_piano/modules/surrogate_synth.py:11-104 (SurrogateAdditive.get_signal, :203-214):
// harmonic_synthesis with the amplitude envelopes multiplied by |decays[t, k]| ** (decay_time[t] U + n % U), t = n / U
// (:76-95), straight from the frame-rate controls.  Same tables, workspace and spans as ddspp_harmonic_synthesis (S = 1);
// decays [R, T, H], decay_time [R, T].
int ddspp_surrogate_harmonic_synthesis(const float* f0_hz, const float* amplitudes, const float* harmonic_distribution,
                                       const float* harmonic_shifts, const float* decays, const float* decay_time,
                                       const float* wlin, const float* whann, float* audio, int R, int T, int H, int U,
                                       float sample_rate, int use_angular_cumsum, int spans, void* workspace,
                                       size_t workspace_bytes, hipStream_t stream) {
    DDSPP_REQUIRE(decays && decay_time, "surrogate_harmonic_synthesis: null decay buffers");
    return harmonic_synthesis_impl(f0_hz, amplitudes, harmonic_distribution, harmonic_shifts, decays, decay_time, wlin, whann,
                                   audio, R, T, 1, H, U, sample_rate, use_angular_cumsum, spans, workspace, workspace_bytes,
                                   stream);
}

// The additive branch of the whole polyphonic group in one call: sum over the P voices of a segment
// of MultiInharmonic.get_signal (inharm_synth.py:272-293) = the `additive/signal` terms of the add
// chain of polyphonic_dag.py:28-37, without the per-voice stems.  Rows are [B * P] (segment major) or,
// with voice_major = 1, [P * B] (the reference Parallelizer's merged layout, sub_modules.py:573-592);
// only oscillators that are audible somewhere in a span get a lane (see osc_count_kernel), so the
// work follows the number of partials below Nyquist instead of P * H.  audio: [B, T * U].
size_t ddspp_polyphonic_additive_workspace_bytes(int B, int P, int T, int S, int H, int U) {
    if (B <= 0 || P <= 0 || T <= 0 || S <= 0 || H <= 0 || U <= 0) return 0;
    const size_t N = (size_t)T * U;
    const size_t nchunks = (N + DDSPP_CHUNK - 1) / DDSPP_CHUNK;
    const size_t V = (size_t)S * H, VP = (V + 63) / 64 * 64 + 64;
    const size_t wmax = (P * V + 63) / 64 + 1;        /* + 1: the last voice's slots start on a slot boundary */
    return (2 * (size_t)B * P * nchunks * VP      /* astart + chunk end phases (worst case: one span per chunk) */
            + (size_t)B * nchunks * (P + 2)       /* nk + wcount */
            + (size_t)B * P                       /* per-row max of the audible counts */
            + (size_t)B * P * nchunks + 64        /* compacted scan: moved flags per (row, chunk), "any" flag */
            + (size_t)B * wmax * N) * 4 + 4096;   /* partial rows */
}

// ... and for every voice's stem (ddspp_polyphonic_stems): the partial rows are one per 32-entry block of the packed list
static size_t stem_blocks_of(int P, int S, int H) { return ((size_t)P * S * ((H + 31) / 32) + 3) & ~(size_t)3; }

size_t ddspp_polyphonic_stems_workspace_bytes(int B, int P, int T, int S, int H, int U) {
    if (B <= 0 || P <= 0 || T <= 0 || S <= 0 || H <= 0 || U <= 0) return 0;
    const size_t N = (size_t)T * U;
    const size_t V = (size_t)S * H, wmax = (P * V + 63) / 64 + 1;
    return ddspp_polyphonic_additive_workspace_bytes(B, P, T, S, H, U) - (size_t)B * wmax * N * 4
           + (size_t)B * stem_blocks_of(P, S, H) * N * 4;
}

static int polyphonic_additive_impl(const float* f0_hz, const float* amplitudes, const float* harmonic_distribution,
                                    const float* harmonic_shifts, const float* inharm_coef, const int* audible,
                                    const float* decays, const float* decay_time,
                                    const float* wlin, const float* whann, const float* phase_state_in, float* audio,
                                    float* audio_last, int B, int P, int T, int S, int H, int U, float sample_rate, int spans,
                                    int voice_major, void* workspace, size_t workspace_bytes, hipStream_t stream,
                                    float* stems = nullptr) {
    DDSPP_REQUIRE(f0_hz && amplitudes && harmonic_distribution && wlin && whann && (audio || stems) && workspace,
                  "polyphonic_additive: null buffer");
    DDSPP_REQUIRE(!stems || (!audio && !audio_last && !decays && !phase_state_in),
                  "polyphonic_stems: stems come without a mix, a decay term or a carried phase state");
    DDSPP_REQUIRE(B > 0 && P > 0 && T > 0 && S > 0 && H > 0 && U > 0, "polyphonic_additive: bad dims");
    DDSPP_REQUIRE(U % BLK == 0, "polyphonic_additive: upsampling=%d must be a multiple of %d", U, BLK);
    DDSPP_REQUIRE(P * S <= 64, "polyphonic_additive: n_synths * n_substrings = %d exceeds 64", P * S);
    DDSPP_REQUIRE((long long)T * U < (1ll << 31) && (T * U) % 4 == 0, "polyphonic_additive: bad sample count");
    const int V = S * H, R = B * P, N = T * U;
    DDSPP_REQUIRE(pick_vpl(V) != 0, "polyphonic_additive: n_substrings*n_harmonics=%d exceeds 512", V);
    DDSPP_REQUIRE(workspace_bytes >= (stems ? ddspp_polyphonic_stems_workspace_bytes(B, P, T, S, H, U)
                                            : ddspp_polyphonic_additive_workspace_bytes(B, P, T, S, H, U)),
                  "polyphonic_additive: workspace too small");
    const int nchunks = (N + DDSPP_CHUNK - 1) / DDSPP_CHUNK;
    // spans: enough (segment, span, slot) tasks to balance 256 CUs
    int sp = spans > 0 ? spans : env_int("DDSPP_OSC_SPANS_COMPACT", 0);
    if (sp <= 0) sp = (env_int("DDSPP_OSC_TARGET_WAVES_COMPACT", 36864) + B * 8 - 1) / (B * 8);
    if (sp > nchunks) sp = nchunks;
    if (sp < 1) sp = 1;
    const int cps = (nchunks + sp - 1) / sp;
    sp = (nchunks + cps - 1) / cps;
    const int vpl_pre = pick_vpl(V);
    const int VP = vpl_pre * 64;
    // A wavefront slot carries 128 audible oscillators (2 per lane); when that leaves the chip short of wavefronts
    // (single segments: B * spans * slots < ~2 per SIMD) it carries 64, twice the wavefronts at half the length.
    // Segments of ONE voice (every voice's stem asked for: the rows of a batch as segments of their own) take 64 as well: a
    // piano row has 40 audible partials on average, a slot of 128 would run half empty (same box, 1024 rows of config 3:
    // 1.90 -> 1.82 ms, two sub-strings 2.79 -> 2.47, every f0 moving 3.80 -> 3.44).
    const int vpl_c = (P == 1 && env_int("DDSPP_OSC_COMPACT_VPL1_SINGLE", 1)) ? 1 :
        ((long long)B * sp * ((P * V + 127) / 128) < env_int("DDSPP_OSC_COMPACT_VPL1_BELOW", 2048)) ? 1 : 2;
    // partial rows (wavefront slots) per segment: with audio_last the last voice's oscillators get slots of their own
    const int split_last = (audio_last && P > 1) ? 1 : 0;
    const int stem_blocks = stems ? (int)stem_blocks_of(P, S, H) : 0;
    const int wmax_a = stems ? (stem_blocks * 32 + 64 * vpl_c - 1) / (64 * vpl_c)
                             : ((P - split_last) * V + 64 * vpl_c - 1) / (64 * vpl_c);
    const int wmax = wmax_a + (split_last ? (V + 64 * vpl_c - 1) / (64 * vpl_c) : 0);

    float* astart = (float*)workspace;
    float* ework = astart + (size_t)R * sp * VP;           // chunk end phases (chunk-parallel pre-pass only)
    int* nk = (int*)(ework + (size_t)R * nchunks * VP);
    int* wcount = nk + (size_t)B * sp * P;
    int* rowmax = wcount + (size_t)B * sp * 2;
    // compacted scan (osc_common.h: scan_tasks): [rowmax R | "some row moves" 1 (+ pad) | moved flags R * npre], all written by
    // osc_count_rows_kernel: nothing to zero
    const int npre_c = sp > 1 ? (sp - 1) * cps : 0;
    // (the sectioned pre-pass -- the one that can leave moving chunks to the scan -- runs on 64-oscillator groups of a row
    // when the row splits into them, span_starts' `split`, or on whole rows of up to 128 oscillators)
    const bool sectioned = (V % 64 == 0 && (long long)R * (V / 64) <= 8192 && !env_int("DDSPP_OSC_PREPASS_WHOLE_ROWS", 0)) ||
                           vpl_pre <= 2;
    const int scan_vpl = sectioned ? 2 : 0;
    const int scan_lanes = 64 * (env_int("DDSPP_OSC_SCAN_VPL", 2) == 1 ? 1 : 2);      // oscillators per scan task
    const int scan_slots = scan_vpl ? (P * V + scan_lanes - 1) / scan_lanes : 0;
    int* scan_ntasks = rowmax + R;
    int* chunk_flag = scan_ntasks + 16;
    float* partial = (float*)(chunk_flag + (size_t)R * npre_c);
    partial = (float*)(((uintptr_t)partial + 255) & ~(uintptr_t)255);

    OscParams p{};
    p.f0 = f0_hz; p.amp = amplitudes; p.hd = harmonic_distribution; p.shifts = harmonic_shifts;
    p.inh = harmonic_shifts ? nullptr : inharm_coef;       // shifts formed in the kernels from the raw inharm_coef
    p.audible = audible;
    p.state_in = phase_state_in;
    p.decays = decays; p.decay_time = decay_time;           // SurrogateAdditive (null: none)
    p.dbg_noflags = env_int("DDSPP_OSC_NO_FLAGS", 0);
    p.wlin = wlin; p.whann = whann;
    p.N = N; p.T = T; p.U = U; p.H = H; p.S = S; p.V = V; p.VP = VP;
    p.spans = sp; p.cps = cps; p.nchunks = nchunks; p.npre = sp > 1 ? (sp - 1) * cps : 0;
    p.sr = sample_rate; p.rsr = 1.0f / sample_rate; p.nyq = sample_rate / 2.0f;
    p.fastdiv = sample_rate_is_checked(sample_rate) && !env_int("DDSPP_NO_FASTDIV", 0);
    p.off_plain = env_int("DDSPP_ANGULAR_OFFSETS_PLAIN", 0) ? 1 : 0;
    p.astart = astart;

    // The chunks in which some frequency moves are scanned by bank_scan_kernel, packed like the bank (round 5): when the
    // per-frame counts and flags of get_controls are there, the sectioned memo pre-pass is the one that runs (same test as
    // launch_memo_prepass), nothing asks for every oscillator's state, and there are chunks to pre-pass at all
    const bool compact_scan = audible && !p.dbg_noflags && scan_vpl && npre_c >= 8 && sp > 1 && !p.state_in &&
                              env_int("DDSPP_OSC_COMPACT_SCAN", 1) && !env_int("DDSPP_OSC_PREPASS_ONE_WAVE", 0) &&
                              !env_int("DDSPP_OSC_PLAIN_PREPASS", 0) && R >= env_int("DDSPP_OSC_MEMO_MIN_WAVES", 256);
    // 1. audible-harmonic counts per (segment, span, voice) (+ with the compacted scan: row maxima and the scan's task list)
    if (compact_scan) {
        static std::atomic<int> call_counter{1};
        const int call_id = call_counter.fetch_add(1) | 0x40000000;
        hipLaunchKernelGGL(osc_count_rows_kernel, dim3(R), dim3(64), 0, stream, audible, nk, R, P, T, U, N, sp, cps, voice_major,
                           rowmax, chunk_flag, scan_ntasks, call_id, npre_c);
        p.rowmax = rowmax;
        p.scan_tasks = chunk_flag; p.scan_ntasks = scan_ntasks; p.scan_slots = scan_slots; p.skip_moving = 1;
        p.scan_call = call_id;
        p.P = P; p.vmajor = voice_major ? 1 : 0;          // (the scan kernel packs the voices of a segment: it reads both)
    }
    // 2. span start offsets for every (row, oscillator) over rows = B * P
    if (sp > 1 || p.state_in) {
        if (audible && T > 1536 && !compact_scan) {   // which 64-oscillator groups are never heard: asked by every wavefront below --
            // of a long row in a kernel of its own (a 3 s row is twelve loads in flight per wavefront: cheaper than a launch)
            hipLaunchKernelGGL(osc_row_max_kernel, dim3(R), dim3(256), 0, stream, audible, rowmax, T);
            p.rowmax = rowmax;
        }
        span_starts(p, R, V, vpl_pre, astart, ework, stream);
    }
    p.skip_moving = 0;
    if (audible && !compact_scan)
        hipLaunchKernelGGL(osc_count_frames_kernel, dim3((R * sp + 255) / 256), dim3(256), 0, stream, audible, nk, R,
                           P, T, U, N, sp, cps, voice_major);
    else if (!audible)
        hipLaunchKernelGGL(osc_count_kernel, dim3((R * sp + 3) / 4), dim3(256), 0, stream, amplitudes,
                           harmonic_distribution, nk, R, P, T, H, U, N, sp, cps, voice_major);
    // 3. the compacted oscillator bank (bank_compact.hip): one wavefront per (segment, span, slot of 64 * vpl audible
    //    oscillators); a piano has about a third of its P * H partials below Nyquist, the slots past the audible set
    //    exit at once (cheap: slot-major workgroup order)
    p.R = B; p.groups = 1; p.vgrp = 64; p.P = P; p.wmax = wmax; p.nslots = wmax; p.vmajor = voice_major ? 1 : 0;
    p.split_last = split_last; p.wmax_a = wmax_a;
    p.half_slots = env_int("DDSPP_OSC_HALF_SLOTS", 1);
    p.pair = (S == 2 && vpl_c == 2 && !decays && env_int("DDSPP_OSC_PAIR", 1)) ? 1 : 0;
    p.held_skip = env_int("DDSPP_OSC_HELD_SKIP", 1);
    p.nk = nk; p.wcount = wcount; p.out = partial;
    if (stems) {                               // every voice's stem: the harmonic sum stops at voice boundaries
        p.pair = 0;
        p.stem_blocks = stem_blocks;
        launch_bank_stems(p, vpl_c, stems, stream);
        DDSPP_LAUNCH_CHECK();
        return DDSPP_OK;
    }
    launch_bank_compact(p, vpl_c, stream);
    // 4. slots -> audio (with audio_last: voices [0, P - 1) -> audio, the last voice -> audio_last)
    if (audio_last && !split_last) {           // P == 1: the only voice is the last one
        launch_bank_slot_sum(p, audio_last, nullptr, stream);
        DDSPP_HIP_CHECK(hipMemsetAsync(audio, 0, (size_t)B * N * sizeof(float), stream));
    } else {
        launch_bank_slot_sum(p, audio, split_last ? audio_last : nullptr, stream);
    }
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

int ddspp_polyphonic_additive(const float* f0_hz, const float* amplitudes, const float* harmonic_distribution,
                              const float* harmonic_shifts, const float* inharm_coef, const int* audible,
                              const float* wlin, const float* whann, const float* phase_state_in, float* audio,
                              float* audio_last, int B, int P, int T, int S, int H, int U, float sample_rate, int spans,
                              int voice_major, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    return polyphonic_additive_impl(f0_hz, amplitudes, harmonic_distribution, harmonic_shifts, inharm_coef, audible, nullptr,
                                    nullptr, wlin, whann, phase_state_in, audio, audio_last, B, P, T, S, H, U, sample_rate,
                                    spans, voice_major, workspace, workspace_bytes, stream);
}

// Every voice's stem of the same call -- stems [B * P, T * U], rows in the order of the controls: what
// synthesize_from_csv.py:99-120 obtains by calling the additive processor once per voice, and what the outputs dictionary of
// default_model.py's node list holds (every `sub_add_i` names a voice's signal).  The compacted bank with the harmonic sum
// stopped at voice boundaries (bank_compact.hip, STEMS): lanes only for audible oscillators, voices packed back to back in
// whole blocks of 32.
int ddspp_polyphonic_stems(const float* f0_hz, const float* amplitudes, const float* harmonic_distribution,
                           const float* harmonic_shifts, const float* inharm_coef, const int* audible, const float* wlin,
                           const float* whann, float* stems, int B, int P, int T, int S, int H, int U, float sample_rate,
                           int spans, int voice_major, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    DDSPP_REQUIRE(stems, "polyphonic_stems: null buffer");
    return polyphonic_additive_impl(f0_hz, amplitudes, harmonic_distribution, harmonic_shifts, inharm_coef, audible, nullptr,
                                    nullptr, wlin, whann, nullptr, nullptr, nullptr, B, P, T, S, H, U, sample_rate, spans,
                                    voice_major, workspace, workspace_bytes, stream, stems);
}

// The same for SurrogateAdditive voices (surrogate_synth.py:11-104, configs/surrogate.gin): every partial's amplitude
// multiplied by |decays[t, k]| ** (decay_time[t] U + n % U) inside the compacted bank.  decays [B * P, T, H] as
// ddspp_surrogate_decays leaves them, decay_time [B * P, T]; one sub-string.
int ddspp_polyphonic_surrogate_additive(const float* f0_hz, const float* amplitudes, const float* harmonic_distribution,
                                        const float* harmonic_shifts, const float* inharm_coef, const int* audible,
                                        const float* decays, const float* decay_time, const float* wlin, const float* whann,
                                        float* audio, float* audio_last, int B, int P, int T, int H, int U, float sample_rate,
                                        int spans, int voice_major, void* workspace, size_t workspace_bytes,
                                        hipStream_t stream) {
    DDSPP_REQUIRE(decays && decay_time, "polyphonic_surrogate_additive: null decay buffers");
    return polyphonic_additive_impl(f0_hz, amplitudes, harmonic_distribution, harmonic_shifts, inharm_coef, audible, decays,
                                    decay_time, wlin, whann, nullptr, audio, audio_last, B, P, T, 1, H, U, sample_rate, spans,
                                    voice_major, workspace, workspace_bytes, stream);
}

// Streaming: the state an oscillator bank carries from one call to the next.  ddsp.core.angular_cumsum restarts the
// phase every 1000 samples and adds the float32 running sum of the chunks' end phases; a call that renders a later
// piece of the same signal therefore needs exactly that sum per (row, oscillator).  phase_state_out[R, V] =
// phase_state_in (or 0) + the end phases of the first n_chunks 1000-sample chunks of these controls, accumulated
// sequentially in float32 -- what a single long call would hold at that point, bit for bit.  T * U >= n_chunks * 1000.
size_t ddspp_oscillator_phase_state_workspace_bytes(int R, int S, int H, int n_chunks) {
    if (R <= 0 || S <= 0 || H <= 0 || n_chunks < 0) return 0;
    const size_t VP = (size_t)pick_vpl(S * H) * 64;
    return 2 * (size_t)R * (n_chunks + 1) * VP * sizeof(float) + 1024;
}

int ddspp_oscillator_phase_state(const float* f0_hz, const float* harmonic_shifts, const float* inharm_coef,
                                 const float* harmonic_distribution, const int* audible, const float* wlin,
                                 const float* phase_state_in, float* phase_state_out, int R, int T, int S, int H, int U,
                                 float sample_rate, int n_chunks, void* workspace, size_t workspace_bytes,
                                 hipStream_t stream) {
    DDSPP_REQUIRE(f0_hz && wlin && phase_state_out && workspace && (harmonic_shifts || inharm_coef || harmonic_distribution),
                  "oscillator_phase_state: null buffer");
    DDSPP_REQUIRE(R > 0 && T > 0 && S > 0 && H > 0 && U > 0 && U % BLK == 0 && n_chunks >= 0, "oscillator_phase_state: bad dims");
    const int V = S * H, N = T * U;
    DDSPP_REQUIRE(pick_vpl(V) != 0, "oscillator_phase_state: n_substrings*n_harmonics=%d exceeds 512", V);
    DDSPP_REQUIRE((long long)n_chunks * DDSPP_CHUNK <= N, "oscillator_phase_state: %d chunks exceed the %d samples given",
                  n_chunks, N);
    DDSPP_REQUIRE(workspace_bytes >= ddspp_oscillator_phase_state_workspace_bytes(R, S, H, n_chunks),
                  "oscillator_phase_state: workspace too small");
    const int vpl_pre = pick_vpl(V), VP = vpl_pre * 64;
    float* astart = (float*)workspace;
    float* ework = astart + (size_t)R * (n_chunks + 1) * VP;
    OscParams p{};
    p.f0 = f0_hz; p.shifts = harmonic_shifts; p.inh = harmonic_shifts ? nullptr : inharm_coef;
    p.hd = harmonic_distribution;      // read (and ignored) only when there are neither shifts nor inharm_coef
    p.audible = audible; p.state_in = phase_state_in;
    p.need_all = 1;          // the state of every oscillator, whether or not this piece lets it be heard
    p.dbg_noflags = env_int("DDSPP_OSC_NO_FLAGS", 0);
    p.wlin = wlin;
    p.N = N; p.T = T; p.U = U; p.H = H; p.S = S; p.V = V; p.VP = VP;
    p.spans = n_chunks + 1; p.cps = 1; p.nchunks = n_chunks + 1; p.npre = n_chunks;
    p.sr = sample_rate; p.rsr = 1.0f / sample_rate; p.nyq = sample_rate / 2.0f;
    p.fastdiv = sample_rate_is_checked(sample_rate) && !env_int("DDSPP_NO_FASTDIV", 0);
    p.off_plain = env_int("DDSPP_ANGULAR_OFFSETS_PLAIN", 0) ? 1 : 0;
    p.astart = astart;
    span_starts(p, R, V, vpl_pre, astart, ework, stream);
    DDSPP_LAUNCH_CHECK();
    DDSPP_HIP_CHECK(hipMemcpy2DAsync(phase_state_out, (size_t)V * sizeof(float), astart + (size_t)n_chunks * VP,
                                     (size_t)(n_chunks + 1) * VP * sizeof(float), (size_t)V * sizeof(float), R,
                                     hipMemcpyDeviceToDevice, stream));
    return DDSPP_OK;
}

}  // extern "C"
#endif  // DDSPP_OSC_PART == 0 (the C-ABI)
