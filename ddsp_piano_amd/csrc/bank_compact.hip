// Compacted polyphonic oscillator bank for gfx950 (CDNA4): the additive branch of a whole segment --
// sum over the P voices of MultiInharmonic.get_signal (ddsp_piano/modules/inharm_synth.py:272-293 ->
// harmonic_synthesis :87-127 -> cos_oscillator_bank :49-84 with ddsp.core.angular_cumsum) -- straight from the
// frame-rate controls, with lanes only for oscillators that are audible somewhere in the span.
//
// This is the kernel the timed step spends most of its time in; it is VALU bound.  What the
// code below is built around, all measured on the MI355X (tools/ubench, profiles/):
//   * a wave64 VALU instruction issues in ~2.3 cycles when its neighbours are independent, ~4.4 when each depends on
//     its predecessor -- other wavefronts of the SIMD do not fill the gap (valu_latency).  So every stage of a block
//     works on all 8 samples x VPL oscillators at once: sixteen independent chains, never one after the other;
//   * a quarter-rate v_cos_f32 costs 8.1 cycles back to back but ~4 cycles extra at every switch between plain and
//     transcendental instructions (trans_overlap): the sixteen cosines of a block are issued as one run;
//   * scheduling barriers between the stages keep the compiler from re-serialising the chains to save registers (the
//     all-purpose osc_kernel template ends up with one chain through two temporaries at the 128-VGPR cap).
// Arithmetic is that of osc_kernel (ddspp_common.h): per (oscillator, sample) the float32 phase scan
// `ph += omega` in the reference's order, `s = ph + off`, an EXACT reduction r = s - rint(s / P) P (one FMA, P =
// float32(2 pi), the reference's modulus), v_cos_f32 on r / P; amplitudes through the Hann cross-fade FMA.
#include <type_traits>

#include "osc_common.h"

namespace ddspp {

namespace {

constexpr float INV_P = 0x1.45f306p-3f;      // RN(1 / (2 pi))

// What a wavefront slot knows before it starts: which (segment, span) it works for, where its region of the packed
// oscillator list lies, and the first oscillator it carries.
struct SlotCtx {
    int lane, row, span, cw_all;
    int n_begin, n_end;
    int qlo, qhi, total, first;       // sub-rows of the region, audible oscillators in it, first oscillator of this slot
};

// The walk of one slot with VPL oscillators per lane (64 VPL oscillators from `first`).  A slot of the VPL = 2 kernel
// that has 64 or fewer oscillators left -- the last slot of a segment half the time, and always the slot(s) of the
// last voice when the caller wants that voice's stem on its own -- runs the VPL = 1 body: half the instructions.
// DECAY (SurrogateAdditive, surrogate_synth.py:76-95): the amplitude of oscillator k in frame t is multiplied by
// |decays[t, k]| ** (decay_time[t] U + r); as in osc_kernel<..., DECAY> the power is evaluated by powf once per frame and
// lane and advances by d ** 8 per block and d per sample.  The next frame's factors are requested one frame ahead.
//
// PAIR (round 5; S = 2, VPL = 2, no decay): the two entries of a lane are the two SUB-STRINGS of one (voice, harmonic)
// instead of two unrelated oscillators.  MultiInharmonic shares amplitudes, harmonic_distribution and harmonic_shifts
// between the sub-strings (inharm_synth.py:279-292: only f0_hz[..., s] differs), so a lane loads them once, forms
// shift_from_inharm once per frame, and -- while both sub-strings sit on the same side of Nyquist, which they do except
// for a partial in the 0.3-cent gap between them -- evaluates ONE Hann cross-fade per sample: a (cos0 + cos1).  The packed
// list then has one entry per (voice, harmonic): a slot carries 64 pairs.
//
// STEMS (round 5; every voice's stem asked for): the same walk, but the harmonic sum stops at voice boundaries.  The packed
// list gives every (voice, sub-string) a whole number of 32-entry blocks (entries past the sub-row's audible count are
// silent lanes), so each half of a wavefront -- and with two oscillators per lane each of the two entries -- belongs to
// ONE voice: the tile keeps the entries apart (rows [0, 16) entry 0, rows [16, 32) entry 1, flushed every 16 samples) and
// a flush writes one row of `out` per 32-entry block instead of one per slot; bank_stems_sum_kernel adds a voice's blocks.
template <int VPL, bool DECAY = false, bool PAIR = false, bool STEMS = false>
__device__ __forceinline__ void bank_slot(const OscParams& p, float* tile, const SlotCtx& c) {
    static_assert(!PAIR || (VPL == 2 && !DECAY), "PAIR: two sub-strings per lane, no decay term");
    static_assert(!STEMS || (!PAIR && !DECAY), "STEMS: plain oscillators only");
    constexpr int TILE_S = (STEMS && VPL == 2) ? TILE / 2 : TILE;        // samples between two flushes
    const int lane = c.lane, row = c.row, span = c.span, cw_all = c.cw_all;
    const int n_begin = c.n_begin, n_end = c.n_end, qlo = c.qlo, qhi = c.qhi, total = c.total;
    const int N = p.N, U = p.U, H = p.H, T = p.T, S = p.S;
    const int SUB = PAIR ? 1 : S;                            // sub-rows of a voice in the packed list
    typedef const __attribute__((address_space(4))) float* cfloat_p;     // wave-uniform tables -> scalar loads
    const cfloat_p wlin_c = (cfloat_p)(uintptr_t)p.wlin;
    const cfloat_p whann_c = (cfloat_p)(uintptr_t)p.whann;
    const float nyq = p.nyq, sr = p.sr, rsr = p.rsr;
    const int* offs = reinterpret_cast<const int*>(tile);    // exclusive offsets of the sub-rows (written by the kernel)
    // The Hann cross-fade weights of a block come from an LDS copy of the window (two broadcast ds_read_b128 per block,
    // issued before the phase scan, used in stage 4): as VECTOR operands.  A VALU instruction with an SGPR operand issues
    // at ~1.7x the cost of one without (tools/ubench/fma_ceiling: 1.75 against 1.03 ns per wave64 multiply-add), and the
    // scalar form needed an s_load_dwordx8 and eight s_mov per block on top.
    // The copy lives in the tile's own padding -- rows are TSTRIDE = 68 words for 64 lanes: four spare words per row, the
    // float4 w[4 q .. 4 q + 3] in row q -- so the kernel's LDS footprint (16 workgroups per CU) does not grow; hops above
    // 4 TILE = 128 (48 kHz) keep the rest behind the tile.
    auto hann4 = [&](int q) {
        // (an arithmetic select: written `q < TILE ? a : b` this was two scalar BRANCHES per call, four per block)
        const int a = q * TSTRIDE + 64, b = TILE * TSTRIDE + 4 * (q - TILE);
        const int off = a + ((b - a) & -(int)(q >= TILE));
        return *reinterpret_cast<const float4*>(tile + off);
    };
    int vk[VPL], vs[VPL], lrow[VPL], vidx[VPL];
    bool valid[VPL];
    float kmul[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int g = c.first + lane + (PAIR ? 0 : 64 * j);
        const int gc = min(g, total - 1);
        int q = qlo;                                         // last sub-row of the region whose offset is <= gc
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (q + step < qhi && offs[q + step] <= gc) q += step;
        int k = gc - offs[q];
        valid[j] = g < total;
        if constexpr (STEMS) {                               // the sub-row's block padding: entries past its audible count
            const int cnt = offs[TSTRIDE + q];
            valid[j] = valid[j] && k < cnt;
            k = min(k, max(cnt - 1, 0));
        }
        lrow[j] = p.vmajor ? (q / SUB) * p.R + row : row * p.P + q / SUB;
        vs[j] = PAIR ? j : q - (q / SUB) * SUB;
        vk[j] = k;
        vidx[j] = vs[j] * H + k;
        kmul[j] = (float)(k + 1);                            // linspace(1, H, H)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- running state ---------------------------------------------------------------------------------------
    float ph[VPL], asum[VPL], off[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        ph[j] = 0.0f;
        asum[j] = 0.0f;
        off[j] = 0.0f;
    }
    if (p.spans > 1 || p.state_in) {       // (a streamed call's first span starts from the carried state)
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            asum[j] = p.astart[((size_t)lrow[j] * p.spans + span) * p.VP + vidx[j]];
            off[j] = chunk_offset(asum[j], p.off_plain);
        }
    }

    // ---- frame controls: x0/a0 = frame t, x1/a1 = frame min(t + 1, T - 1); the raw values of the frame after that
    // are requested one whole frame early (q_*), so their latency hides behind U samples of arithmetic.
    //   hf(t, v) = (f0[t, s] * k) * (1 + shift[t, k])     inharm_synth.py:106-108
    //   ha(t, v) = amp[t] * hd[t, k]                      inharm_synth.py:112
    float x0[VPL], x1[VPL], a0[VPL], a1[VPL];
    float q_f0[VPL], q_sh[VPL], q_hd[VPL], q_amp[VPL];
    // Round 6: with get_controls' per-frame counts at hand (p.audible: 1 + the last harmonic whose amp * hd is not zero) a
    // harmonic at or above its frame's count is taken as silent FROM THE COUNT -- the product it replaces is exactly zero --
    // so get_controls need not write that part of harmonic_distribution at all (InharmParams::hd_sparse: two thirds of the
    // [R, T, H] tensor at a piano's note mix; what the load returns there is never used).  One more 4-byte load per lane and
    // frame; the compare takes the place of the `valid` select.
    const bool by_count = p.audible != nullptr;
    int q_cnt[VPL], vkv[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        vkv[j] = valid[j] ? vk[j] : 0x7fffffff;
        q_cnt[j] = 0x10000;
    }
    // DECAY: d0 = |decays| of the current frame, e_blk = d0 ** (decay_time U + r) at the start of the current block,
    // d0_8 = d0 ** 8; nd / ndt = the next frame's raw factor and time (requested when the current frame starts)
    float d0[DECAY ? VPL : 1], e_blk[DECAY ? VPL : 1], d0_8[DECAY ? VPL : 1], nd[DECAY ? VPL : 1], ndt[DECAY ? VPL : 1];
    auto decay_request = [&](int tt) {
        if constexpr (DECAY) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const size_t fr = (size_t)lrow[j] * T + tt;
                nd[j] = p.decays[fr * H + vk[j]];
                ndt[j] = p.decay_time[fr];
            }
        }
    };
    auto decay_start = [&](int r0) {       // the requested frame becomes the current one, at its sample r0
        if constexpr (DECAY) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                d0[j] = fabsf(nd[j]);
                e_blk[j] = powf(d0[j], ndt[j] * (float)U + (float)r0);
                const float d2 = d0[j] * d0[j], d4 = d2 * d2;
                d0_8[j] = d4 * d4;
            }
        }
    };
    const bool has_shifts = p.shifts != nullptr, from_inh = !has_shifts && p.inh != nullptr;
    auto frame_request = [&](int tt) {
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const size_t fr = (size_t)lrow[j] * T + tt;
            q_f0[j] = p.f0[fr * S + vs[j]];
            if (PAIR && j > 0) {                   // the sub-strings share everything but f0_hz
                q_amp[j] = q_amp[0];
                q_sh[j] = q_sh[0];
                q_hd[j] = q_hd[0];
                q_cnt[j] = q_cnt[0];
                continue;
            }
            q_amp[j] = p.amp[fr];
            if (by_count) q_cnt[j] = p.audible[fr] & 0xffff;
            q_sh[j] = has_shifts ? p.shifts[fr * H + vk[j]] : (from_inh ? p.inh[fr] : 0.0f);
            q_hd[j] = p.hd[fr * H + vk[j]];
        }
    };
    float r_f0[VPL], r_sh[VPL];                    // the raw values x1 was formed from (held-note test at a frame boundary)
    auto frame_finish = [&](float* xf, float* xa) {
        float one_plus_shift = 1.0f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            r_f0[j] = q_f0[j];
            r_sh[j] = q_sh[j];
            float f = q_f0[j] * kmul[j];
            if (has_shifts) f = f * (1.0f + q_sh[j]);
            else if (from_inh) {                                                         // get_inharmonic_freq, per lane and frame
                if (!(PAIR && j > 0)) one_plus_shift = 1.0f + shift_from_inharm(q_sh[j], kmul[j]);
                f = f * one_plus_shift;
            }
            const float a = q_amp[j] * q_hd[j];
            xf[j] = valid[j] ? f : 0.0f;
            xa[j] = vkv[j] < q_cnt[j] ? a : 0.0f;
        }
    };
    // per-frame classification (wave-uniform):
    //   fast       every frequency of the frame pair is >= 0, either 0 or comfortably normal, and small enough that a
    //              chunk's phase stays below 2^22 * 2 pi  -> the exact constant division (div_const) and the one-FMA
    //              2 pi reduction are valid (ddspp_common.h)
    //   const_freq x0 == x1 in every lane (a held note): fe == x0 exactly, omega is the per-frame constant om_c
    //   need_mask  some oscillator crosses Nyquist inside the frame pair -> per-sample remove_above_nyquist
    bool fast = false, const_freq = false, need_mask = true;
    float om_c[VPL], da[VPL], am0[VPL];    // am0 / da: the frame pair's amplitudes as the blocks use them
    const float f_big = 3900.0f * sr;              // 1008 samples of omega(f_big) stay below 2.6e7 rad
    auto classify_frame = [&]() {
        bool ok = true, msk = false, cst = true;
        bool gone_j[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const float lo = fminf(x0[j], x1[j]), hi = fmaxf(x0[j], x1[j]);
            ok = ok && (lo > 1e-28f || (lo == 0.0f && (hi == 0.0f || hi > 1e-24f))) && (hi < f_big);
            // above Nyquist for the whole frame pair: masked once, here (a0 / a1 themselves stay as they are: the
            // next pair starts from the unmasked a1)
            const bool gone = lo >= nyq;
            gone_j[j] = gone;
            am0[j] = gone ? 0.0f : a0[j];
            da[j] = gone ? 0.0f : a1[j] - a0[j];
            msk = msk || (lo < nyq && hi >= nyq);
            cst = cst && (x0[j] == x1[j]);
            om_c[j] = omega_of<false>(x0[j], sr, rsr);
        }
        // PAIR: the shared cross-fade needs both sub-strings of every lane on the same side of Nyquist for the whole frame
        // pair; a frame where they are not (a partial inside the detune gap) takes the masked moving path, which keeps
        // the amplitudes apart (x0 + 0 * w = x0: the same phases, whatever the frame's frequencies do)
        if (PAIR) msk = msk || (gone_j[0] != gone_j[VPL - 1]);
        fast = __all(ok) && p.fastdiv;
        need_mask = __any(msk);
        const_freq = __all(cst) && !(PAIR && need_mask);
    };

    int t = n_begin / U, r = n_begin - t * U;
    frame_request(t);
    frame_finish(x0, a0);
    frame_request(min(t + 1, T - 1));
    frame_finish(x1, a1);
    frame_request(min(t + 2, T - 1));
    classify_frame();
    decay_request(t);
    decay_start(r);
    decay_request(min(t + 1, T - 1));

    float* out_row = STEMS ? p.out + ((size_t)row * p.stem_blocks + (c.first >> 5)) * N
                           : p.out + ((size_t)row * p.wmax + cw_all) * N;
    int cpos = 0, tpos = 0, tile_n0 = n_begin;

    auto flush_tile = [&](int nt0, int count) {
#if defined(DDSPP_BANK_ABLATE) && (DDSPP_BANK_ABLATE & 1)
        return;
#endif
        if constexpr (STEMS) {
            // lane (col, entry, half) adds the 32 lane partials of sample `col` that belong to block 2 entry + half of the slot
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int col = lane & (TILE_S - 1), ent = VPL == 2 ? (lane >> 4) & 1 : 0, half = lane >> 5;
            const float4* src = reinterpret_cast<const float4*>(tile + (ent * TILE_S + col) * TSTRIDE + half * 32);
            typedef float f4v __attribute__((ext_vector_type(4)));
            f4v tv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) tv[i] = *reinterpret_cast<const f4v*>(src + i);
            asm volatile("" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tv[4]), "+v"(tv[5]), "+v"(tv[6]), "+v"(tv[7]));
            float4 s4 = make_float4(tv[0].x, tv[0].y, tv[0].z, tv[0].w);
#pragma unroll
            for (int i = 1; i < 8; ++i) {
                s4.x += tv[i].x; s4.y += tv[i].y; s4.z += tv[i].z; s4.w += tv[i].w;
            }
            const float s = (s4.x + s4.y) + (s4.z + s4.w);
            const int blk = 2 * ent + half;
            if (col < count && c.first + 32 * blk < total) out_row[(size_t)blk * N + nt0 + col] = s;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            return;
        }
        // column sums: lane (col, half) adds 32 of the 64 lane partials of sample `col`
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int col = lane & 31, half = lane >> 5;
        const float4* src = reinterpret_cast<const float4*>(tile + col * TSTRIDE + half * 32);
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v tv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) tv[i] = *reinterpret_cast<const f4v*>(src + i);
        asm volatile("" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(tv[4]), "+v"(tv[5]), "+v"(tv[6]), "+v"(tv[7]));
        float4 s4 = make_float4(tv[0].x, tv[0].y, tv[0].z, tv[0].w);
#pragma unroll
        for (int i = 1; i < 8; ++i) {
            s4.x += tv[i].x; s4.y += tv[i].y; s4.z += tv[i].z; s4.w += tv[i].w;
        }
        float s = (s4.x + s4.y) + (s4.z + s4.w);
        s += __shfl_xor(s, 32);
        if (lane < count) out_row[nt0 + lane] = s;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // stages 2-4 of a block whose phases pv[i][j] (before the chunk offset) are known; MASK: per-sample Nyquist mask
    auto finish_block = [&](float (*pv)[VPL], const float (*fe)[VPL], const float* w1, auto mask_tag) {
        constexpr bool MASK = decltype(mask_tag)::value;
        __builtin_amdgcn_sched_barrier(0);
        // ---- stage 2: s = phase + offsets (the reference's float32 add), then s - q P with ONE multiple q of the
        // reference's modulus P = float32(2 pi) per oscillator and block: q = rint(s_first / P).  The phase advances by
        // less than pi per sample, so r = s - q P lies in [-pi, pi + 7 omega]; the FMA forms it from the exact
        // product: no rounding while |r| < 16 (both terms are multiples of 2^-21), at most 2^-20 rad for the few
        // partials near Nyquist late in a block.  v_cos_f32 takes revolutions and folds whole turns exactly, so
        // cos(r / 2 pi) is cos(floormod(s, P)) to ~1e-6 rad -- two instructions per sample less than a per-sample q.
        float q0[VPL];
#pragma unroll
        for (int i = 0; i < BLK; ++i)
#pragma unroll
            for (int j = 0; j < VPL; ++j) pv[i][j] = pv[i][j] + off[j];
#pragma unroll
        for (int j = 0; j < VPL; ++j) q0[j] = -__builtin_rintf(pv[0][j] * INV_P);
#pragma unroll
        for (int i = 0; i < BLK; ++i)
#pragma unroll
            for (int j = 0; j < VPL; ++j) pv[i][j] = __builtin_fmaf(q0[j], DDSPP_TWO_PI_F32, pv[i][j]);
#pragma unroll
        for (int i = 0; i < BLK; ++i)
#pragma unroll
            for (int j = 0; j < VPL; ++j) pv[i][j] = pv[i][j] * INV_P;
        __builtin_amdgcn_sched_barrier(0);
        // ---- stage 3: the cosines of the block in one run -------------------------------------------------------
#pragma unroll
        for (int i = 0; i < BLK; ++i)
#pragma unroll
#if defined(DDSPP_BANK_ABLATE) && (DDSPP_BANK_ABLATE & 4)
            for (int j = 0; j < VPL; ++j) pv[i][j] = pv[i][j] * 0.5f;
#else
            for (int j = 0; j < VPL; ++j) pv[i][j] = __builtin_amdgcn_cosf(pv[i][j]);
#endif
        __builtin_amdgcn_sched_barrier(0);
        // ---- stage 4: Hann cross-fade of the amplitudes (core.upsample_with_windows: a0 w[U + r] + a1 w[r] with
        // w[U + r] + w[r] = 1 to an ulp = a0 + (a1 - a0) w[r]), Nyquist mask, harmonic sum over the lane's own -------
        float acc[BLK];
        float er[DECAY ? VPL : 1];         // DECAY: the power at sample i of the block
        if constexpr (DECAY) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) er[j] = e_blk[j];
        }
        if constexpr (STEMS && VPL == 2) {
            // the two entries of a lane may belong to two voices: kept apart in the tile
#pragma unroll
            for (int j = 0; j < VPL; ++j)
#pragma unroll
                for (int i = 0; i < BLK; ++i) {
                    float a = __builtin_fmaf(da[j], w1[i], am0[j]);
                    if (MASK) a = (fe[i][j] >= nyq) ? 0.0f : a;
                    tile[(j * TILE_S + tpos + i) * TSTRIDE + lane] = a * pv[i][j];
                }
            return;
        }
        if constexpr (PAIR && !MASK) {
            // the two sub-strings of a (voice, harmonic) under ONE cross-fade (classify_frame: same side of Nyquist)
#pragma unroll
            for (int i = 0; i < BLK; ++i) {
                const float a = __builtin_fmaf(da[0], w1[i], am0[0]);
                acc[i] = a * (pv[i][0] + pv[i][1]);
            }
        } else {
#pragma unroll
        for (int i = 0; i < BLK; ++i) {
            float a = __builtin_fmaf(da[0], w1[i], am0[0]);
            if constexpr (DECAY) {
                a = a * er[0];                                             // surrogate_synth.py:91-95
                er[0] = er[0] * d0[0];
            }
            if (MASK) a = (fe[i][0] >= nyq) ? 0.0f : a;                    // remove_above_nyquist
            acc[i] = a * pv[i][0];
        }
#pragma unroll
        for (int j = 1; j < VPL; ++j)
#pragma unroll
            for (int i = 0; i < BLK; ++i) {
                float a = __builtin_fmaf(da[j], w1[i], am0[j]);
                if constexpr (DECAY) {
                    a = a * er[j];
                    er[j] = er[j] * d0[j];
                }
                if (MASK) a = (fe[i][j] >= nyq) ? 0.0f : a;
                acc[i] = __builtin_fmaf(a, pv[i][j], acc[i]);
            }
        }
        if constexpr (DECAY) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) e_blk[j] = e_blk[j] * d0_8[j];
        }
#if defined(DDSPP_BANK_ABLATE) && (DDSPP_BANK_ABLATE & 2)
        float keep = 0.f;
#pragma unroll
        for (int i = 0; i < BLK; ++i) keep += acc[i];
        if (keep == 1.2345e30f) tile[lane] = keep;
#else
#pragma unroll
        for (int i = 0; i < BLK; ++i) tile[(tpos + i) * TSTRIDE + lane] = acc[i];
#endif
    };

    // The path a block takes -- constant frequency, moving, moving with a Nyquist mask, generic -- is a property of the
    // FRAME pair (classify_frame), so the block loop exists once per path (round 4: as one loop with the choice inside,
    // every block paid two or three taken scalar branches to reach its body, and the constant-frequency path paid for the
    // moving paths' scalar weight prefetch).  `block_tail`: what follows every block whatever its path.
    auto hann_weights = [&](float* w1) {
        const float4 wa = hann4(r >> 2), wb = hann4((r >> 2) + 1);
        w1[0] = wa.x; w1[1] = wa.y; w1[2] = wa.z; w1[3] = wa.w;
        w1[4] = wb.x; w1[5] = wb.y; w1[6] = wb.z; w1[7] = wb.w;
    };
    auto block_tail = [&](int n0) {
        // ---- tile bookkeeping ---------------------------------------------------------------------------------
        tpos += BLK;
        if (tpos == TILE_S || n0 + BLK >= n_end) {
            flush_tile(tile_n0, tpos);
            tile_n0 += tpos;
            tpos = 0;
        }
        // ---- chunk boundary (ddsp.core.angular_cumsum) ---------------------------------------------------------
        cpos += BLK;
        if (cpos == DDSPP_CHUNK) {
            cpos = 0;
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const float e = mod_2pi(ph[j]);      // phase[:, :, -1] % 2pi
                asum[j] = asum[j] + e;               // cumsum over chunks (float32, sequential)
                off[j] = chunk_offset(asum[j], p.off_plain);           // % 2pi
                ph[j] = 0.0f;
            }
        }
        r += BLK;
    };
    // bilinear weights wlin[n0 + i] of a block (moving paths only): scalar loads issued one block ahead (wave-uniform
    // addresses -> s_load_dwordx8); the first block of a moving frame loads its own
    auto moving_blocks = [&](int& n0, int nf_end, auto mask_tag) {
        float wl[BLK];
#pragma unroll
        for (int i = 0; i < BLK; ++i) wl[i] = wlin_c[n0 + i];
        for (; n0 < nf_end; n0 += BLK) {
            float wlnext[BLK], w1[BLK];
            {
                const int nn = min(n0 + BLK, N - BLK);
#pragma unroll
                for (int i = 0; i < BLK; ++i) wlnext[i] = wlin_c[nn + i];
                hann_weights(w1);
            }
            float pv[BLK][VPL], fe[BLK][VPL], om[BLK][VPL];
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const float dx = x1[j] - x0[j];
#pragma unroll
                for (int i = 0; i < BLK; ++i) fe[i][j] = x0[j] + dx * wl[i];          // legacy bilinear (core.resample)
            }
            if (r + BLK == U && wl[BLK - 1] == WALK_NEXT_ROW) {      // a long file: see osc_common.h (wave-uniform)
#pragma unroll
                for (int i = 0; i < BLK; ++i)
#pragma unroll
                    for (int j = 0; j < VPL; ++j) fe[i][j] = (wl[i] == WALK_NEXT_ROW) ? x1[j] : fe[i][j];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < BLK; ++i)
#pragma unroll
                for (int j = 0; j < VPL; ++j) om[i][j] = omega_of<true>(fe[i][j], sr, rsr);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < BLK; ++i)
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    ph[j] = ph[j] + om[i][j];
                    pv[i][j] = ph[j];
                }
            finish_block(pv, fe, w1, mask_tag);
#pragma unroll
            for (int i = 0; i < BLK; ++i) wl[i] = wlnext[i];
            block_tail(n0);
        }
    };

    // Two loops: the outer one walks frames, the inner one the blocks of a frame.  The controls of frame t + 2 are
    // requested when frame t starts and only touched when it ends: inside the inner loop nothing depends on them, so
    // no wait for them (and no register shuffling of loop-carried copies) sits between two blocks.
    for (int n0 = n_begin; n0 < n_end;) {
    const int nf_end = min(n0 + (U - r), n_end);
    if (fast && const_freq) {
        // (also tried: the four blocks of a whole tile back to back with no test between them -- 2 % slower, the code grows)
        for (; n0 < nf_end; n0 += BLK) {
            float w1[BLK];
            hann_weights(w1);
            // ---- stage 1: the float32 phase scan (VPL sequential chains, interleaved) ---------------------------
            float pv[BLK][VPL];
#pragma unroll
            for (int i = 0; i < BLK; ++i)
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    ph[j] = ph[j] + om_c[j];
                    pv[i][j] = ph[j];
                }
            finish_block(pv, pv, w1, std::false_type{});
            // (a marker the paths do not share: without one the compiler sinks the common stages of the inlined
            // finish_block bodies into one copy behind selector flags -- a dozen scalar branches per block)
            asm volatile("; bank block: constant frequency");
            block_tail(n0);
        }
    } else if (fast && need_mask) {
        moving_blocks(n0, nf_end, std::true_type{});
        asm volatile("; bank blocks: moving frequency, Nyquist mask");
    } else if (fast) {
        moving_blocks(n0, nf_end, std::false_type{});
        asm volatile("; bank blocks: moving frequency");
    } else {
        for (; n0 < nf_end; n0 += BLK) {
            // generic path (negative / denormal / huge frequencies, unchecked sample rates): IEEE division, fmod
            // based floormod, one sample at a time -- correctness only
#pragma unroll 1
            for (int i = 0; i < BLK; ++i) {
                const float wli = wlin_c[n0 + i], whi = whann_c[r + i];
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < VPL; ++j) {
                    const float f = (wli == WALK_NEXT_ROW) ? x1[j] : x0[j] + (x1[j] - x0[j]) * wli;
                    ph[j] = ph[j] + omega_of<false>(f, sr, rsr);
                    float a = __builtin_fmaf(da[j], whi, am0[j]);
                    if constexpr (DECAY) {
                        a = a * e_blk[j];
                        e_blk[j] = e_blk[j] * d0[j];
                    }
                    a = (f >= nyq) ? 0.0f : a;
                    if constexpr (STEMS && VPL == 2) {
                        tile[(j * TILE_S + tpos + i) * TSTRIDE + lane] = a * cos_reduced(mod_2pi(ph[j] + off[j]));
                        continue;
                    }
                    acc = __builtin_fmaf(a, cos_reduced(mod_2pi(ph[j] + off[j])), acc);
                }
                if constexpr (!(STEMS && VPL == 2)) tile[(tpos + i) * TSTRIDE + lane] = acc;
            }
            block_tail(n0);
        }
    }
    // ---- frame boundary ----------------------------------------------------------------------------------------
    if (r == U) {
        r = 0;
        ++t;
        // A held note: the frame pair before was constant (x0 == x1 in every lane) and the next frame's raw f0_hz and
        // inharm_coef (or shifts) are the very same numbers -> the new pair is the old one.  Only the amplitudes move:
        // no per-lane square root (shift_from_inharm), no classification, om_c / fast / need_mask stay.
        decay_start(0);                              // (DECAY) the frame requested one frame ago becomes the current one
        decay_request(min(t + 1, T - 1));
        bool same = const_freq;
#pragma unroll
        for (int j = 0; j < VPL; ++j) same = same && (q_f0[j] == r_f0[j]) && (q_sh[j] == r_sh[j]);
        if (__all(same) && p.held_skip) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                a0[j] = a1[j];
                const float a = q_amp[j] * q_hd[j];
                a1[j] = vkv[j] < q_cnt[j] ? a : 0.0f;
                const bool gone = x0[j] >= nyq;
                am0[j] = gone ? 0.0f : a0[j];
                da[j] = gone ? 0.0f : a1[j] - a0[j];
            }
            frame_request(min(t + 2, T - 1));
        } else {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                x0[j] = x1[j];
                a0[j] = a1[j];
            }
            frame_finish(x1, a1);                    // raw values requested one frame ago
            frame_request(min(t + 2, T - 1));
            classify_frame();
        }
    }
    }
}

// One wavefront per workgroup: slots past the audible set exit at once and give their place to the next workgroup.
template <int VPL, bool DECAY = false, bool PAIR = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
bank_compact_kernel(const OscParams p) {
    extern __shared__ float lds_dyn[];
    float* tile = lds_dyn;                                   // [TILE][TSTRIDE]
    SlotCtx c;
    c.lane = threadIdx.x & 63;
    // workgroup index = slot major, (segment, span) minor: consecutive workgroups go round-robin to the 8 XCDs, so
    // every XCD gets the same mix of busy (low slots) and idle workgroups and the busy ones are dispatched first
    const int nbs = p.R * p.spans;
    c.cw_all = blockIdx.x / nbs;
    const int bs = blockIdx.x - c.cw_all * nbs;
    c.row = bs / p.spans;                                    // segment b
    c.span = bs - c.row * p.spans;
    const int c0 = c.span * p.cps, c1 = min(c0 + p.cps, p.nchunks);
    c.n_begin = c0 * DDSPP_CHUNK;
    c.n_end = min(c1 * DDSPP_CHUNK, p.N);
    const int lane = c.lane, S = PAIR ? 1 : p.S;             // (PAIR: one packed entry per (voice, harmonic), both sub-strings)
    constexpr int CAP = PAIR ? 64 : 64 * VPL;                // packed entries a slot carries

    // ---- which oscillators this slot carries ------------------------------------------------------------------
    // The audible oscillators of the segment's (voice, sub-string) rows are packed back to back: sub-row q
    // contributes its first nk harmonics.  Region A = voices [0, P - split_last), region B = the last voice.
    const int Q = p.P * S, Qa = (p.P - p.split_last) * S;
    const bool in_b = c.cw_all >= p.wmax_a;
    const int cw = in_b ? c.cw_all - p.wmax_a : c.cw_all;
    const int len = lane < Q ? p.nk[((size_t)c.row * p.spans + c.span) * p.P + lane / S] : 0;
    int incl = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const int total_a = Qa > 0 ? __shfl(incl, Qa - 1) : 0;
    const int total_b = __shfl(incl, 63) - total_a;
    if (c.cw_all == 0 && lane == 0) {
        int* wc = p.wcount + ((size_t)c.row * p.spans + c.span) * 2;
        wc[0] = (total_a + CAP - 1) / CAP;
        wc[1] = (total_b + CAP - 1) / CAP;
    }
    c.total = in_b ? total_b : total_a;
    c.first = CAP * cw;
    if (c.first >= c.total) return;                          // nothing audible left for this slot
    c.qlo = in_b ? Qa : 0;
    c.qhi = in_b ? Q : Qa;
    const int base = in_b ? total_a : 0;
    int* offs = reinterpret_cast<int*>(tile);                // exclusive offsets of the sub-rows, via LDS
    offs[lane] = incl - len - base;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < p.U; i += 64) {                   // w[0 .. U) of the Hann window into the tile's padding (bank_slot)
        const int q = i >> 2;
        (q < TILE ? tile + q * TSTRIDE + 64 : tile + TILE * TSTRIDE + 4 * (q - TILE))[i & 3] = p.whann[i];
    }
    if constexpr (PAIR)
        bank_slot<2, false, true>(p, tile, c);
    else if (VPL == 2 && (c.total - c.first > 64 || !p.half_slots))
        bank_slot<2, DECAY>(p, tile, c);
    else
        bank_slot<1, DECAY>(p, tile, c);
}

// audio[b, n] = sum over the wavefront slots that were used, in slot order (deterministic); with split_last the last
// voice's slots go to audio_last and the others' to audio
__global__ void __launch_bounds__(256) bank_slot_sum_kernel(const float* __restrict__ partial, const int* __restrict__ wcount,
                                                          float* __restrict__ out, float* __restrict__ out_last, int B,
                                                          int N, int wmax, int wmax_a, int spans, int cps) {
    const int n4 = N / 4;
    const size_t total = (size_t)B * n4;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int b = (int)(g / n4), i = (int)(g - (size_t)b * n4);
        const int span = min((4 * i) / (cps * DDSPP_CHUNK), spans - 1);
        const int wa = wcount[((size_t)b * spans + span) * 2], wb = wcount[((size_t)b * spans + span) * 2 + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), accb = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < wa; ++w) {
            const float4 v = reinterpret_cast<const float4*>(partial + ((size_t)b * wmax + w) * N)[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        for (int w = 0; w < wb; ++w) {
            const float4 v = reinterpret_cast<const float4*>(partial + ((size_t)b * wmax + wmax_a + w) * N)[i];
            accb.x += v.x; accb.y += v.y; accb.z += v.z; accb.w += v.w;
        }
        if (out_last) {
            reinterpret_cast<float4*>(out_last + (size_t)b * N)[i] = accb;
        } else {
            acc.x += accb.x; acc.y += accb.y; acc.z += accb.z; acc.w += accb.w;
        }
        reinterpret_cast<float4*>(out + (size_t)b * N)[i] = acc;
    }
}

// Every voice's stem (bank_slot<.., STEMS>): the packed list of a segment with every (voice, sub-string) sub-row rounded up
// to whole 32-entry blocks; a slot writes one row of p.out per block it carries: p.out [B, stem_blocks, N].
template <int VPL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4)))
bank_stems_kernel(const OscParams p) {
    extern __shared__ float lds_dyn[];
    float* tile = lds_dyn;
    SlotCtx c;
    c.lane = threadIdx.x & 63;
    const int nbs = p.R * p.spans;
    c.cw_all = blockIdx.x / nbs;
    const int bs = blockIdx.x - c.cw_all * nbs;
    c.row = bs / p.spans;
    c.span = bs - c.row * p.spans;
    const int c0 = c.span * p.cps, c1 = min(c0 + p.cps, p.nchunks);
    c.n_begin = c0 * DDSPP_CHUNK;
    c.n_end = min(c1 * DDSPP_CHUNK, p.N);
    const int lane = c.lane, S = p.S, Q = p.P * S;
    constexpr int CAP = 64 * VPL;
    const int cnt = lane < Q ? p.nk[((size_t)c.row * p.spans + c.span) * p.P + lane / S] : 0;
    const int len = (cnt + 31) & ~31;
    int incl = len;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    c.total = __shfl(incl, 63);
    c.first = CAP * c.cw_all;
    if (c.first >= c.total) return;
    c.qlo = 0;
    c.qhi = Q;
    int* offs = reinterpret_cast<int*>(tile);
    offs[lane] = incl - len;                                 // exclusive block-aligned offsets of the sub-rows
    offs[TSTRIDE + lane] = cnt;                              // ... and their audible counts (bank_slot: silent padding lanes)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < p.U; i += 64) {
        const int q = i >> 2;
        (q < TILE ? tile + q * TSTRIDE + 64 : tile + TILE * TSTRIDE + 4 * (q - TILE))[i & 3] = p.whann[i];
    }
    if (VPL == 2 && (c.total - c.first > 64 || !p.half_slots))
        bank_slot<2, false, false, true>(p, tile, c);
    else
        bank_slot<1, false, false, true>(p, tile, c);
}

// stems[row of (b, v), n] = the blocks of voice v in the packed list of (b, span of n), in block order (deterministic);
// a voice that is silent in the span gets zeros.  One workgroup per (segment, span, voice).
__global__ void __launch_bounds__(256) bank_stems_sum_kernel(const float* __restrict__ blocks, const int* __restrict__ nk,
                                                           float* __restrict__ stems, int B, int P, int S, int N,
                                                           int stem_blocks, int spans, int cps, int vmajor) {
    int id = blockIdx.x;
    const int v = id % P; id /= P;
    const int span = id % spans;
    const int b = id / spans;
    const int* cnt = nk + ((size_t)b * spans + span) * P;
    int first = 0;
    for (int u = 0; u < v; ++u) first += S * ((cnt[u] + 31) >> 5);
    const int nblk = S * ((cnt[v] + 31) >> 5);
    const int n0 = span * cps * DDSPP_CHUNK, n1 = min(n0 + cps * DDSPP_CHUNK, N);
    const float* src = blocks + ((size_t)b * stem_blocks + first) * N;
    float* dst = stems + (size_t)(vmajor ? v * B + b : b * P + v) * N;
    for (int i = n0 / 4 + threadIdx.x; i < n1 / 4; i += 256) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int w = 0; w < nblk; ++w) {
            const float4 x = reinterpret_cast<const float4*>(src + (size_t)w * N)[i];
            acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
        }
        reinterpret_cast<float4*>(dst)[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Compacted scan of the chunks whose frequencies move (round 5; OscParams::scan_tasks).  The span starts of the bank are
// running sums of chunk END PHASES e[c] = (sum of the chunk's 1000 omegas, sequentially in float32) % 2 pi of every chunk
// before the span (ddsp.core.angular_cumsum).  osc_prepass_fused_kernel memoises the chunks of held notes; a chunk in
// which some frequency of the segment moves is listed by osc_count_frames_kernel and scanned HERE, sample by sample, with
// the bank's packing: the oscillators of a segment that are audible anywhere in their row, voices back to back, 64 VPL to
// a wavefront -- instead of one wavefront per (row, 64 harmonics), which at a piano's note mix has 1.7 lanes per audible
// partial.  A persistent grid walks (slot, segment, chunk); which chunks are flagged is only known on the device: none --
// held notes -- and the wavefronts leave at once.  Same arithmetic, same order, same bits as the pre-pass's own moving branch
// (scan_block_staged); a (row, chunk) that is constant is the pre-pass's and left out of the packed list here.
#ifndef SCAN_WPE
#define SCAN_WPE 4      // wavefronts per SIMD of bank_scan_kernel (an even count: a SIMD issues for two wavefronts at a time)
#endif
constexpr int SCAN_WQ = (DDSPP_CHUNK + 255) / 256;          // float4 loads per lane that cover a chunk's weights
template <int VPL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SCAN_WPE, SCAN_WPE)))
bank_scan_kernel(const OscParams p) {
    extern __shared__ float lds_dyn[];
    int* offs = reinterpret_cast<int*>(lds_dyn);             // [64] exclusive offsets of the sub-rows of the segment
    float* wlds = lds_dyn + 64;                              // [256 SCAN_WQ + 8] interpolation weights of the chunk being scanned
    const int lane = threadIdx.x & 63;
    if (wave_uniform(*p.scan_ntasks) != p.scan_call) return;      // held notes: no row moves anywhere, nothing to scan
    const int S = p.S, H = p.H, T = p.T, U = p.U, N = p.N, Q = p.P * S;
    const bool has_shifts = p.shifts != nullptr, from_inh = !has_shifts && p.inh != nullptr;
    const float srv = in_vgpr(p.sr), rsrv = in_vgpr(p.rsr);
    // No atomics anywhere (a first version appended tasks to a list and handed them out with one counter: 70 000 atomics on
    // one address cost more than the scan).  A task is (slot, segment, chunk), slot-major: a wavefront's tasks t = w, w + G,
    // ... run through the slots, so every wavefront gets its share of the full low slots and of the empty high ones.
    const int nentries = p.R * p.npre;
    for (int task = blockIdx.x; task < nentries * p.scan_slots; task += gridDim.x) {
        const int slot = task / nentries, bc = task - slot * nentries;
        const int seg = bc / p.npre, c = bc - seg * p.npre;
        // ---- the packed list of the segment: sub-row q (voice, sub-string) contributes its first rowmax harmonics ----
        int len = 0, mv = 0;
        if (lane < Q) {
            const int v = lane / S;
            const int vrow = p.vmajor ? v * p.R + seg : seg * p.P + v;
            mv = p.scan_tasks[(size_t)vrow * p.npre + c];
            // only the voices that MOVE in this chunk are packed (round 6): the pre-pass memoises every (row, chunk) that is
            // constant, whatever the segment's other voices do.  Before, one moving voice had the whole segment's audible
            // oscillators scanned: with note onsets and releases in every chunk of a performance (bench.py midi_like: 80 changes
            // per 3 s segment over 72 chunks) that was every chunk of every voice -- 0.71 ms, more than the bank itself.
            len = mv ? p.rowmax[vrow] : 0;
        }
        if (!__any(mv != 0)) continue;                       // no frequency of this segment moves in this chunk: the pre-pass has it
        int incl = len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        const int total = __shfl(incl, 63);
        const int first = 64 * VPL * slot;
        if (first >= total) continue;                        // (wave-uniform)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        offs[lane] = incl - len;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int vk[VPL], vs[VPL], lrow[VPL], vidx[VPL];
        bool valid[VPL];
        float kmul[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int g = first + lane + 64 * j;
            const int gc = min(g, total - 1);
            int q = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (q + step < Q && offs[q + step] <= gc) q += step;
            const int k = gc - offs[q];
            lrow[j] = p.vmajor ? (q / S) * p.R + seg : seg * p.P + q / S;
            vs[j] = q - (q / S) * S;
            vk[j] = k;
            vidx[j] = vs[j] * H + k;
            valid[j] = g < total;
            kmul[j] = (float)(k + 1);
        }
        auto hf_raw = [&](int tt, float* rf, float* rs) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                const size_t fr = (size_t)lrow[j] * T + tt;
                rf[j] = p.f0[fr * S + vs[j]];
                rs[j] = has_shifts ? p.shifts[fr * H + vk[j]] : (from_inh ? p.inh[fr] : 0.0f);
            }
        };
        auto hf_calc = [&](const float* rf, const float* rs, float* xf) {
#pragma unroll
            for (int j = 0; j < VPL; ++j) {
                float f = rf[j] * kmul[j];
                if (has_shifts) f = f * (1.0f + rs[j]);
                else if (from_inh) f = f * (1.0f + shift_from_inharm(rs[j], kmul[j]));
                xf[j] = valid[j] ? f : 0.0f;
            }
        };
        // ---- the chunk, frame by frame (osc_prepass_fused_kernel's moving branch) ----------------------------------
        const int n_lo = c * DDSPP_CHUNK, n_hi = min(n_lo + DDSPP_CHUNK, N);
        int tt = n_lo / U, r = n_lo - tt * U;
        float ph[VPL], x0[VPL], x1[VPL], qf[VPL], qs[VPL];
#pragma unroll
        for (int j = 0; j < VPL; ++j) ph[j] = 0.0f;
        {
            float rf[VPL], rs[VPL];
            hf_raw(tt, rf, rs);
            hf_calc(rf, rs, x0);
            hf_raw(min(tt + 1, T - 1), rf, rs);
            hf_calc(rf, rs, x1);
        }
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
        // the interpolation weights of the WHOLE chunk go through LDS at once (1000 floats: sixteen per lane, all loads in
        // flight together): one memory latency per task -- staged frame by frame, as the pre-pass does with twice the
        // wavefronts per SIMD, the scan waited for eleven of them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            float4 wq[SCAN_WQ];
#pragma unroll
            for (int u = 0; u < SCAN_WQ; ++u) {
                const int idx = 4 * (lane + 64 * u);
                wq[u] = (n_lo + idx < n_hi) ? *reinterpret_cast<const float4*>(p.wlin + n_lo + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < SCAN_WQ; ++u) *reinterpret_cast<float4*>(wlds + 4 * (lane + 64 * u)) = wq[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float wl[BLK], wn[BLK];
        auto weights_at = [&](int off, float* w) {
            const float4 wa = *reinterpret_cast<const float4*>(wlds + off);
            const float4 wb = *reinterpret_cast<const float4*>(wlds + off + 4);
            w[0] = wa.x; w[1] = wa.y; w[2] = wa.z; w[3] = wa.w;
            w[4] = wb.x; w[5] = wb.y; w[6] = wb.z; w[7] = wb.w;
        };
        for (int n = n_lo; n < n_hi;) {
            const int nf = min(n + (U - r), n_hi);
            weights_at(n - n_lo, wl);
            auto blocks = [&](auto fast_tag) {
                constexpr bool FAST = decltype(fast_tag)::value;
                int woff = n - n_lo;
                for (; n + BLK < nf; n += BLK) {
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
        for (int j = 0; j < VPL; ++j)
            if (valid[j]) p.echunk[((size_t)lrow[j] * p.npre + c) * p.VP + vidx[j]] = mod_2pi(ph[j]);
    }
}

}  // namespace

void launch_bank_compact(const OscParams& p, int vpl, hipStream_t stream) {
    const size_t lds = (size_t)(TILE * TSTRIDE + (p.U > 4 * TILE ? p.U - 4 * TILE : 0)) * sizeof(float);   // tile (+ what of the Hann window its padding cannot hold)
    const dim3 grid((unsigned)((size_t)p.R * p.spans * p.wmax)), blk(64);
    if (p.decays) {                        // SurrogateAdditive: the decay term rides in the block's amplitude stage
        if (vpl == 1) hipLaunchKernelGGL((bank_compact_kernel<1, true>), grid, blk, lds, stream, p);
        else hipLaunchKernelGGL((bank_compact_kernel<2, true>), grid, blk, lds, stream, p);
        return;
    }
    if (vpl == 1) hipLaunchKernelGGL((bank_compact_kernel<1>), grid, blk, lds, stream, p);
    else if (p.pair) hipLaunchKernelGGL((bank_compact_kernel<2, false, true>), grid, blk, lds, stream, p);
    else hipLaunchKernelGGL((bank_compact_kernel<2>), grid, blk, lds, stream, p);
}

void launch_bank_stems(const OscParams& p, int vpl, float* stems, hipStream_t stream) {
    const size_t lds = (size_t)(TILE * TSTRIDE + (p.U > 4 * TILE ? p.U - 4 * TILE : 0)) * sizeof(float);
    const dim3 grid((unsigned)((size_t)p.R * p.spans * p.wmax)), blk(64);
    if (vpl == 1) hipLaunchKernelGGL((bank_stems_kernel<1>), grid, blk, lds, stream, p);
    else hipLaunchKernelGGL((bank_stems_kernel<2>), grid, blk, lds, stream, p);
    hipLaunchKernelGGL(bank_stems_sum_kernel, dim3((unsigned)((size_t)p.R * p.spans * p.P)), dim3(256), 0, stream, p.out, p.nk,
                       stems, p.R, p.P, p.S, p.N, p.stem_blocks, p.spans, p.cps, p.vmajor);
}

void launch_bank_scan(const OscParams& p, int vpl, hipStream_t stream) {
    // persistent grid: as many one-wavefront workgroups as the chip holds a few times over; the task count lives on the device
    const size_t lds = (size_t)(64 + 256 * SCAN_WQ + 8) * sizeof(float);
    const unsigned grid = (unsigned)ddspp_option_literal("DDSPP_OSC_SCAN_WAVES", 8192);
    if (vpl == 1) hipLaunchKernelGGL((bank_scan_kernel<1>), dim3(grid), dim3(64), lds, stream, p);
    else hipLaunchKernelGGL((bank_scan_kernel<2>), dim3(grid), dim3(64), lds, stream, p);
}

void launch_bank_slot_sum(const OscParams& p, float* audio, float* audio_last, hipStream_t stream) {
    size_t blocks = ((size_t)p.R * (p.N / 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bank_slot_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p.out, p.wcount, audio, audio_last,
                       p.R, p.N, p.wmax, p.wmax_a, p.spans, p.cps);
}

}  // namespace ddspp
