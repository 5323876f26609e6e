// cos_oscillator_bank (ddsp_piano/modules/inharm_synth.py:49-84) on MATERIALISED envelopes, summed, angular cumsum, a whole
// number of 64-harmonic groups per row: the HBM-roofline kernel of the path (8 bytes read per
// oscillator-sample, 4 written per audio sample), written for that one job (round 6).  oscillator.hip's osc_kernel stays the
// route of every other shape / flag combination and is what this kernel is bit-compared with (tests/test_gpu_osc.py).
//
// Same decomposition as osc_kernel -- workgroup = (row, span), wavefront g = harmonics 64 g .. 64 g + 63, lane = harmonic, time
// walked sequentially (the reference's scan order), [32 samples][64 lanes] LDS tile for the harmonic sum (same order of
// additions: the audio is bitwise what osc_kernel writes) -- and what is different:
//   * non-temporal loads with immediate offsets from ONE scalar block address (no per-load address arithmetic; the
//     envelopes are read once, nothing of them should stay in L2 / MALL: 6.35 -> 6.88 TB/s for the bare pattern,
//     profiles/r06_ubench.txt)
//   * block classification in 2 integer ops per sample (max / min over the bit patterns) instead of 6 float ops
//   * no per-lane validity selects (the route requires H % 64 == 0), no amplitude-activity vote
#include "osc_common.h"

namespace ddspp {

constexpr int STREAM_OB = 0;       // the library's choice of OB (see osc_stream_kernel)
constexpr int STREAM_NB = 4;       // register ring: blocks of BLK samples; STREAM_NB - 1 in flight while one is consumed

template <bool FDIV, bool FMOD>
__device__ __forceinline__ void stream_block(const float (&fb)[BLK], const float (&ab)[BLK], float& ph, float off, float sr, float rsr,
                                             float nyq, float* __restrict__ tile_col) {
    float om[BLK], pv[BLK];
#pragma unroll
    for (int i = 0; i < BLK; ++i) om[i] = omega_of<FDIV>(fb[i], sr, rsr);          // inharm_synth.py:69-70
#pragma unroll
    for (int i = 0; i < BLK; ++i) {                                                // sequential float32 scan
        ph = ph + om[i];
        pv[i] = ph;
    }
#pragma unroll
    for (int i = 0; i < BLK; ++i) {
        const float a = (fb[i] >= nyq) ? 0.0f : ab[i];                             // remove_above_nyquist  :65-67
        const float s = pv[i] + off;                                               // phase + offsets
        const float c = FMOD ? cos_of_phase_fast(s) : cos_reduced(mod_2pi(s));     // % 2pi ; cos
        tile_col[i * TSTRIDE] = __builtin_fmaf(a, c, 0.0f);
    }
}

// ABL: timing experiments only (tools/ubench/osc_graded.hip instantiates them; the library only ABL = 0):
//   bit 0 = the ring of loads and a sum of what arrives, bit 1 = no tile flush / barrier / store (wrong audio),
//   bit 2 = the flush without its global store, bit 3 = the flush without its s_barrier,
//   bit 4 = every row's audio stored over the same 4 KB again and again (the stores stay in L2: issue cost without HBM writes),
//   bit 5 = an extra wavefront per workgroup does the stores (launch with 64 (G + 1) threads)
// OB: audio samples gathered in LDS before they are stored (0 = every tile's 32 samples stored as they are summed);
// NTS: the audio is stored non-temporally.  What the stores cost (profiles/r06_ubench.txt, tools/ubench/store_cost.hip):
// 0.3 GB of audio beside 75.5 GB of reads take 0.5 ms of a 11.1 ms read stream with nt / sc1 stores and 1.0 ms with plain
// ones (write-allocate), whether 128 bytes or 4 KB are stored at a time, from this wavefront or from a third one, to the
// audio buffer or to the same 4 KB again and again: proportional to the bytes written and to nothing else, so OB = 0.
template <int G, int ABL = 0, int OB = 0, bool NTS = false>
__global__ void __launch_bounds__(64 * (G + ((ABL & 32) ? 1 : 0))) osc_stream_kernel(const OscParams p) {
    extern __shared__ float lds_dyn[];
    const int lane = threadIdx.x & 63;
    const int grp = wave_uniform(threadIdx.x >> 6);
    const int row = blockIdx.x / p.spans;
    const int span = blockIdx.x - row * p.spans;
    const int c0 = span * p.cps, c1 = min(c0 + p.cps, p.nchunks);
    const int N = p.N;
    constexpr int H = 64 * G;
    const int n_begin = c0 * DDSPP_CHUNK;
    const int n_end = min(c1 * DDSPP_CHUNK, N);            // both multiples of BLK (N % BLK == 0 is required by the entry point)
    float* tile = lds_dyn + grp * (TILE * TSTRIDE);
    float* comb = lds_dyn + G * (TILE * TSTRIDE);          // [2][G][32] combine buffer
    float* obuf = comb + 2 * G * 32;                       // [OB] audio waiting to be stored (wavefront 0 only)
    int comb_buf = 0, opos = 0;
    const float sr = p.sr, rsr = p.rsr, nyq = p.nyq;

    float ph = 0.0f, asum = 0.0f, off = 0.0f;
    if (p.spans > 1) {
        asum = p.astart[((size_t)row * p.spans + span) * p.VP + grp * 64 + lane];
        off = chunk_offset(asum, p.off_plain);
    }
    // the fast forms (div_const, cos_of_phase_fast) need 0 <= s < 2^22 * 2pi: phases restart in every chunk, so
    // f < 3000 sr bounds them by 1.9e7, and the offset (wrapped: < 2pi; plain: the running sum) by off_ok
    bool off_ok = __all(off < 1.0e6f);

    const float* fe_row = p.fe + (size_t)row * N * H;
    const float* ae_row = p.ae + (size_t)row * N * H;
    float* out_row = p.out + (size_t)row * N;
    const int col = grp * 64 + lane;
    const int last_block = n_end - BLK;

    if constexpr ((ABL & 32) != 0) {           // experiment: an extra wavefront does nothing but the combine + store of every tile
        if (grp == G) {
            int nt0 = n_begin, buf = 0;
            while (nt0 < n_end) {
                __syncthreads();
                const int count = min(32, n_end - nt0);
                const float* cb = comb + buf * (G * 32);
                if (lane < count) {
                    float tot = cb[lane];
#pragma unroll
                    for (int g = 1; g < G; ++g) tot += cb[g * 32 + lane];
                    if (NTS) __builtin_nontemporal_store(tot, out_row + nt0 + lane);
                    else out_row[nt0 + lane] = tot;
                }
                buf ^= 1;
                nt0 += count;
            }
            return;
        }
    }
    float fbuf[STREAM_NB][BLK], abuf[STREAM_NB][BLK];
    auto load_block = [&](int n0, float (&fb)[BLK], float (&ab)[BLK]) {
        const int nc = min(n0, last_block);                // past the span: the last block again (a wave-uniform address, no branch)
        const float* fs = fe_row + (size_t)nc * H;
        const float* as = ae_row + (size_t)nc * H;
#pragma unroll
        for (int i = 0; i < BLK; ++i) {
            fb[i] = __builtin_nontemporal_load(fs + i * H + col);
            ab[i] = __builtin_nontemporal_load(as + i * H + col);
        }
    };

    const uint32_t hi_lim = __float_as_uint(fminf(3000.0f * sr, 3.0e38f));
    const uint32_t lo_lim = __float_as_uint(1e-28f) - 1u;

    auto flush_tile = [&](int nt0, int count) {
        // column sums in osc_kernel's order: lane (col, half) adds 32 of the 64 lane partials of sample `col`
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int tcol = lane & 31, half = lane >> 5;
        const float4* src = reinterpret_cast<const float4*>(tile + tcol * TSTRIDE + half * 32);
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
        float tot = s;
        if (G > 1) {
            float* cb = comb + comb_buf * (G * 32);
            if (lane < 32) cb[grp * 32 + lane] = s;
            if constexpr ((ABL & 8) == 0) __syncthreads();               // also what keeps the row's wavefronts at the same place of the stream
            if ((ABL & 32) == 0 && grp == 0 && lane < 32) {
                tot = cb[lane];
#pragma unroll
                for (int g = 1; g < G; ++g) tot += cb[g * 32 + lane];
            }
            comb_buf ^= 1;
        }
        if constexpr ((ABL & 32) != 0) {
        } else if (OB == 0) {
            if (grp == 0 && lane < count) {
                if constexpr ((ABL & 4) != 0) {
                    if (tot == 1.2345e30f) out_row[nt0 + lane] = tot;
                } else if constexpr ((ABL & 16) != 0) {
                    if (NTS) __builtin_nontemporal_store(tot, out_row + (nt0 & 1023) + lane);
                    else out_row[(nt0 & 1023) + lane] = tot;
                } else if (NTS) __builtin_nontemporal_store(tot, out_row + nt0 + lane);
                else out_row[nt0 + lane] = tot;
            }
        } else if (grp == 0) {
            if (lane < count) obuf[opos + lane] = tot;
            opos += count;
            if (opos == OB || nt0 + count >= n_end) {                    // whole 16-byte pieces: count is a multiple of BLK
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                typedef float f4v __attribute__((ext_vector_type(4)));
                f4v* dst = reinterpret_cast<f4v*>(out_row + (nt0 + count - opos));
                for (int i = lane; i * 4 < opos; i += 64) {
                    const f4v v = *reinterpret_cast<const f4v*>(obuf + i * 4);
                    if (NTS) __builtin_nontemporal_store(v, dst + i);
                    else dst[i] = v;
                }
                opos = 0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    int tpos = 0, tile_n0 = n_begin, cpos = 0;
    auto do_block = [&](int n0, const float (&fb)[BLK], const float (&ab)[BLK]) {
        if constexpr ((ABL & 1) != 0) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < BLK; ++i) s += fb[i] + ab[i];
            if (s == 1.2345e30f) p.out[0] = s;
            return;
        }
        // every frequency of the block is 0 or in [1e-28, min(3000 sr, 3e38)): as unsigned integers non-negative floats
        // order like their values, negatives / NaN / inf are above every bound, and u - 1 sends 0 to the top
        uint32_t hi = 0u, lo = 0xffffffffu;
#pragma unroll
        for (int i = 0; i < BLK; ++i) {
            const uint32_t u = __float_as_uint(fb[i]);
            hi = max(hi, u);
            lo = min(lo, u - 1u);
        }
        const bool fast = __all(hi < hi_lim && lo >= lo_lim) && off_ok;
        float* tcolp = tile + tpos * TSTRIDE + lane;
        if (fast && p.fastdiv) stream_block<true, true>(fb, ab, ph, off, sr, rsr, nyq, tcolp);
        else if (fast) stream_block<false, true>(fb, ab, ph, off, sr, rsr, nyq, tcolp);
        else stream_block<false, false>(fb, ab, ph, off, sr, rsr, nyq, tcolp);
        if constexpr ((ABL & 2) == 0) {
            tpos += BLK;
            if (tpos == TILE || n0 + BLK >= n_end) {
                flush_tile(tile_n0, tpos);
                tile_n0 += tpos;
                tpos = 0;
            }
        }
        // chunk boundary (ddsp.core.angular_cumsum)
        cpos += BLK;
        if (cpos == DDSPP_CHUNK) {
            cpos = 0;
            const float e = mod_2pi(ph);               // phase[:, :, -1] % 2pi
            asum = asum + e;                           // cumsum over chunks (float32, sequential)
            off = chunk_offset(asum, p.off_plain);     // % 2pi
            ph = 0.0f;
            off_ok = __all(off < 1.0e6f);
        }
    };

#pragma unroll
    for (int b = 0; b < STREAM_NB - 1; ++b) load_block(n_begin + b * BLK, fbuf[b], abuf[b]);
    // whole tiles: NO branch around a load anywhere in this loop -- hipcc's s_waitcnt bookkeeping joins the states of every
    // path into a block, and a path that skips sixteen loads makes every later wait assume they were never issued
    // (`if (nb0 < n_end)` around the ring positions, as osc_kernel has it, turned the counted waits into vmcnt(14): one
    // block in flight instead of three)
    int n0 = n_begin;
    for (; n0 + STREAM_NB * BLK <= n_end; n0 += STREAM_NB * BLK) {
#pragma unroll
        for (int b = 0; b < STREAM_NB; ++b) {
            load_block(n0 + (b + STREAM_NB - 1) * BLK, fbuf[(b + STREAM_NB - 1) % STREAM_NB], abuf[(b + STREAM_NB - 1) % STREAM_NB]);
            do_block(n0 + b * BLK, fbuf[b], abuf[b]);
        }
    }
    // the last, partial tile: its blocks are already in the ring (positions 0 ..)
#pragma unroll
    for (int b = 0; b < STREAM_NB - 1; ++b)
        if (n0 + b * BLK < n_end) do_block(n0 + b * BLK, fbuf[b], abuf[b]);
}

// p as ddspp_cos_oscillator_bank fills it (materialised source, angular, summed); the plan's groups / vgrp are not used:
// a row always gets H / 64 wavefronts here (astart is indexed by oscillator, whatever wrote it)
bool osc_stream_applies(const OscParams& p) {
    return p.fe && p.ae && p.H % 64 == 0 && p.H / 64 <= 4 && p.N % BLK == 0 && p.N >= BLK;
}

void launch_osc_stream(const OscParams& p, hipStream_t stream) {
    const int G = p.H / 64;
    const dim3 grid((unsigned)(p.R * p.spans)), blk(64 * G);
    const size_t lds = ((size_t)G * (TILE * TSTRIDE) + 2 * G * 32 + STREAM_OB) * sizeof(float);
    switch (G) {
        case 1: hipLaunchKernelGGL((osc_stream_kernel<1, 0, STREAM_OB, true>), grid, blk, lds, stream, p); break;
        case 2: hipLaunchKernelGGL((osc_stream_kernel<2, 0, STREAM_OB, true>), grid, blk, lds, stream, p); break;
        case 3: hipLaunchKernelGGL((osc_stream_kernel<3, 0, STREAM_OB, true>), grid, blk, lds, stream, p); break;
        default: hipLaunchKernelGGL((osc_stream_kernel<4, 0, STREAM_OB, true>), grid, blk, lds, stream, p); break;
    }
}

}  // namespace ddspp
