// Time-varying FilteredNoise FIR for gfx950.
//
// Replaces ddsp.core.frequency_filter -> frequency_impulse_response -> fft_convolve as reached from
// DynamicSizeFilteredNoise.get_signal (ddsp_piano/modules/filtered_noise_synth.py:27-42):
//   ir_t   = window * irfft(magnitudes[t])  (zero phase -> causal linear phase, Lw taps)
//   z[m]   = sum_j noise[j] * ir_{j / U}[m - j]           (each U-sample block filtered by its own
//                                                          frame's FIR, tails overlap-added)
//   out[n] = z[n + delay],  delay = (Lw - 1) // 2 - 1     (crop_and_compensate_delay, 'same')
// The reference evaluates the block convolutions with 512-point FFTs; for Lw <= 254 taps and
// U <= 192 samples the direct form is cheaper on CDNA4 than three FFTs per frame plus their HBM
// round trips, and it equals the FFT result to float32 round-off.
//
// Kernels
//   fir_eo_mfma_kernel : the FIR design as two GEMMs on the matrix cores (v_mfma_f32_16x16x4_f32, exact f32)
//       through the even/odd split of the inverse real DFT; fir_eo_kernel is its VALU form (K = 128).
//   fir_from_magnitudes_kernel : ir[r, t, :] = magnitudes[r, t, :] @ M, M = the (windowed,
//       shifted) inverse real DFT matrix [K, Lw] built by the host in float64.  Lane = tap, the
//       tap's column of M lives in registers, the frame's magnitudes arrive through scalar loads.
//   tv_fir_kernel : gather form, one workgroup per 2048 (or 1024) output samples of a row.  The frame FIRs
//       and the noise window are staged in LDS once per workgroup; a lane owns 16 consecutive outputs and
//       a quarter of the input blocks that reach them: 64 FMAs per six conflict-free ds_read_b128, the four
//       partial sums meet in registers (two __shfl_xor rounds).
//   tv_fir_generic_kernel : thread-per-output fallback for shapes the tiled kernel does not take.
//   uniform_noise_kernel : Philox4x32-10 counter based U(-1, 1) noise (the reference draws an
//       unseeded tf.random.uniform; parity is defined with the noise tensor supplied).
#include "ddspp_common.h"
#include "noise_win.h"

namespace ddspp {

// ------------------------------------------------------------------------------------------------
// impulse responses from magnitudes
// ------------------------------------------------------------------------------------------------
// Lane j computes the unique tap uniq[j] (and writes its mirror image when the FIR is symmetric:
// the zero-phase -> linear-phase construction makes ir[c + m] == ir[c - m]).  The K magnitudes of a
// frame are wave-uniform: they arrive through scalar loads, 16 at a time, and feed the FMAs as SGPR
// operands; the tap's column of M stays in K VGPRs for the whole run of frames.
template <int K, int THREADS>
__global__ void __launch_bounds__(THREADS) fir_from_magnitudes_kernel(const float* __restrict__ mags,
                                                                    const float* __restrict__ M,
                                                                    const int* __restrict__ uniq,
                                                                    const int* __restrict__ mirror,
                                                                    float* __restrict__ ir, int frames,
                                                                    int Lw, int n_uniq,
                                                                    int frames_per_block) {
    const int j = threadIdx.x;
    const bool active = j < n_uniq;
    const int tap = active ? uniq[j] : 0;
    const int mir = active ? mirror[j] : -1;
    float col[K];
#pragma unroll
    for (int k = 0; k < K; ++k) col[k] = active ? M[(size_t)k * Lw + tap] : 0.0f;
    const int f0 = blockIdx.x * frames_per_block;
    const int f1 = min(f0 + frames_per_block, frames);
    for (int f = f0; f < f1; ++f) {
        const float* __restrict__ mg = mags + (size_t)f * K;       // wave-uniform address
        float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll
        for (int k = 0; k < K; k += 2) {
            acc0 = __builtin_fmaf(mg[k], col[k], acc0);
            acc1 = __builtin_fmaf(mg[k + 1], col[k + 1], acc1);
        }
        const float acc = acc0 + acc1;
        if (active) {
            ir[(size_t)f * Lw + tap] = acc;
            if (mir >= 0) ir[(size_t)f * Lw + mir] = acc;
        }
    }
}

// Even/odd split of the inverse real DFT (full-window case, Lh = 2 (K - 1), half = K - 1):
//   cos(2 pi (half - j) k / Lh) = (-1)^k cos(2 pi j k / Lh)
//   E[j] = sum_{k even} c_k m_k cos(.)/Lh,  O[j] = sum_{k odd} ...,   z[j] = E + O,  z[half - j] = E - O
// so lane j (j <= half / 2) yields up to four taps of the symmetric FIR from K FMAs -- a quarter of the
// dense product.  The frame's magnitudes are staged in LDS (coalesced) and read back as wave-uniform
// 16-byte blocks; the lane's two table columns stay in K registers.
template <int KH>   // KH = K / 2 (K even)
__global__ void __launch_bounds__(256) fir_eo_kernel(const float* __restrict__ mags,     // [frames, 2 KH]
                                                   const float* __restrict__ CE,       // [KH, NJ]
                                                   const float* __restrict__ CO,       // [KH, NJ]
                                                   const int* __restrict__ tap_idx,    // [NJ, 4]
                                                   const float* __restrict__ tap_we,   // [NJ, 4]
                                                   const float* __restrict__ tap_wo,   // [NJ, 4]
                                                   float* __restrict__ ir, int frames, int Lw, int NJ,
                                                   int frames_per_block, float bias, ScaleFn scale) {
    extern __shared__ __attribute__((aligned(16))) float mtile[];       // [frames_per_block][2 KH], then the output rows
    constexpr int K = 2 * KH;
    float* otile = mtile + frames_per_block * K;                        // [frames_per_block][Lw]
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int j = min(lane, NJ - 1);
    const bool active = lane < NJ;
    float ce[KH], co[KH];
#pragma unroll
    for (int q = 0; q < KH; ++q) {
        ce[q] = CE[q * NJ + j];
        co[q] = CO[q * NJ + j];
    }
    int tidx[4];
    float twe[4], two[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        tidx[s] = active ? tap_idx[j * 4 + s] : -1;
        twe[s] = tap_we[j * 4 + s];
        two[s] = tap_wo[j * 4 + s];
    }
    // A workgroup walks several tiles of frames_per_block frames: the 2 KH table registers loaded above (as many
    // bytes as a tile itself) are paid once per workgroup, not once per tile.
    const int ntiles = (frames + frames_per_block - 1) / frames_per_block;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int f0 = tile * frames_per_block;
        const int nf = min(frames_per_block, frames - f0);
        {   // coalesced copy of the tile
            const float4* src = reinterpret_cast<const float4*>(mags + (size_t)f0 * K);
            float4* dst = reinterpret_cast<float4*>(mtile);
            const int n4 = nf * (K / 4);
            // raw network outputs: FilteredNoise.get_controls' scale_fn(magnitudes + initial_bias) on the way in
            with_scale_kind(scale.kind, [&](auto kind) {
                for (int i = threadIdx.x; i < n4; i += 256) dst[i] = scale4_of<decltype(kind)::value>(scale, src[i], bias);
            });
        }
        __syncthreads();
        for (int f = wib; f < nf; f += 4) {
            const float4* mg = reinterpret_cast<const float4*>(mtile + f * K);   // wave-uniform address
            float e0 = 0.f, e1 = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int q = 0; q < KH; q += 2) {
                const float4 m = mg[q / 2];                  // magnitudes 2q, 2q+1, 2q+2, 2q+3
                e0 = __builtin_fmaf(m.x, ce[q], e0);
                o0 = __builtin_fmaf(m.y, co[q], o0);
                e1 = __builtin_fmaf(m.z, ce[q + 1], e1);
                o1 = __builtin_fmaf(m.w, co[q + 1], o1);
            }
            const float E = e0 + e1, O = o0 + o1;
            float* dst = otile + f * Lw;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (tidx[s] >= 0) dst[tidx[s]] = __builtin_fmaf(two[s], O, twe[s] * E);
        }
        __syncthreads();                 // rows complete; the magnitude tile may be overwritten by the next one
        {   // the tile's impulse responses are one contiguous block of ir: full 16-byte stores instead of four
            // 4-byte scatters per lane and frame (partial-line writes held the kernel at 2.9 TB/s)
            float* dst = ir + (size_t)f0 * Lw;                       // 16-byte aligned: frames_per_block * Lw % 4 == 0
            const int nfl = nf * Lw, n4 = nfl / 4;
            float4* dst4 = reinterpret_cast<float4*>(dst);
            const float4* src4 = reinterpret_cast<const float4*>(otile);
            for (int i = threadIdx.x; i < n4; i += 256) dst4[i] = src4[i];
            for (int i = 4 * n4 + threadIdx.x; i < nfl; i += 256) dst[i] = otile[i];
        }
    }
}

// The same even/odd product on the matrix cores.  E = M_even @ CE and O = M_odd @ CO are [frames, K/2] x [K/2, NJ]
// GEMMs: a wavefront owns 16 frames, v_mfma_f32_16x16x4_f32 accumulates the 16 x 16 blocks of E and O over K/8
// steps (exact f32, a k-ordered fma chain).  The table fragments (B operands) live in registers for the whole
// kernel; the A operands are one ds_read_b64 per step (magnitudes 2k, 2k + 1 of the lane's frame) shared by all
// the 16-column blocks -- 2 * JT matrix instructions (2048 * JT MACs) per LDS read instead of 4 FMAs.  The rows
// of the tile are assembled in LDS (tap weights applied on the way) and leave as full 16-byte stores.
// Wavefronts are independent (private LDS region, no workgroup barrier) and walk tiles of 16 frames.
template <int KH, int JT>   // KH = K / 2, JT = ceil(NJ / 16)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) fir_eo_mfma_kernel(const float* __restrict__ mags,     // [frames, 2 KH]
                                                        const float* __restrict__ CE,       // [KH, NJ]
                                                        const float* __restrict__ CO,       // [KH, NJ]
                                                        const int* __restrict__ tap_idx,    // [NJ, 4]
                                                        const float* __restrict__ tap_we,   // [NJ, 4]
                                                        const float* __restrict__ tap_wo,   // [NJ, 4]
                                                        float* __restrict__ ir, int frames, int Lw, int NJ,
                                                        int region_floats, float bias, ScaleFn scale) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    constexpr int K = 2 * KH, KS = KH / 4, FT = 16;      // FT frames per tile
    constexpr int ASTR = K + 4;                            // padded row stride of the magnitude tile (16-byte aligned)
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    float* reg = lds_dyn + (size_t)wib * region_floats;    // this wavefront's LDS: magnitudes, then the output rows
    const int col = lane & 15, kq = lane >> 4;

    // B fragments: lane holds CE/CO[4 s + kq][16 jt + col]
    float bE[JT][KS], bO[JT][KS];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        const int j = min(16 * jt + col, NJ - 1);
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            bE[jt][st] = CE[(4 * st + kq) * NJ + j];
            bO[jt][st] = CO[(4 * st + kq) * NJ + j];
        }
    }
    // tap tables (index, even weight, odd weight per lane column and slot): shared by the workgroup, in LDS
    int* tix = reinterpret_cast<int*>(lds_dyn + (size_t)4 * region_floats);     // [16 JT][4]
    float* twe = reinterpret_cast<float*>(tix + 64 * JT);
    float* two = twe + 64 * JT;
    for (int i = threadIdx.x; i < 64 * JT; i += 256) {
        const int j = i >> 2;
        tix[i] = j < NJ ? tap_idx[i] : -1;
        twe[i] = j < NJ ? tap_we[i] : 0.0f;
        two[i] = j < NJ ? tap_wo[i] : 0.0f;
    }
    __syncthreads();
    const int ntiles = (frames + FT - 1) / FT;
    for (int tile = blockIdx.x * 4 + wib; tile < ntiles; tile += gridDim.x * 4) {
        const int f0 = tile * FT;
        const int nf = min(FT, frames - f0);
        // ---- magnitudes of the 16 frames -> LDS (scale_fn on the way when they are raw network outputs)
        {
            constexpr int PER_ROW = K / 4;                          // float4 per frame
            constexpr int N4 = FT * PER_ROW;
            const float4* src = reinterpret_cast<const float4*>(mags + (size_t)f0 * K);
            const int lim = nf * PER_ROW - 1;
            with_scale_kind(scale.kind, [&](auto kind) {
#pragma unroll
                for (int u = 0; u < (N4 + 63) / 64; ++u) {
                    const int i = lane + 64 * u;
                    if (i < N4) {
                        const float4 m = scale4_of<decltype(kind)::value>(scale, src[min(i, lim)], bias);
                        const int fr = i / PER_ROW, c4 = i - fr * PER_ROW;
                        *reinterpret_cast<float4*>(reg + fr * ASTR + 4 * c4) = m;
                    }
                }
            });
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- E, O accumulation: A fragment = magnitudes (2 kk, 2 kk + 1), kk = 4 s + kq, of frame `col`
        f32x4 accE[JT], accO[JT];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            accE[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
            accO[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const float* arow = reg + col * ASTR + 2 * kq;
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            const float2 a = *reinterpret_cast<const float2*>(arow + 8 * st);
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) {
                accE[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bE[jt][st], accE[jt], 0, 0, 0);
                accO[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bO[jt][st], accO[jt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                 // every lane is done with the magnitude tile: reuse the region
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- D[row = 4 kq + r][col]: frame 4 kq + r, lane column j = 16 jt + col -> its (up to) four taps
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float E = accE[jt][r], O = accO[jt][r];
                float* dst = reg + (4 * kq + r) * Lw;
                const int4 ti = *reinterpret_cast<const int4*>(tix + 4 * (16 * jt + col));
                const float4 we = *reinterpret_cast<const float4*>(twe + 4 * (16 * jt + col));
                const float4 wo = *reinterpret_cast<const float4*>(two + 4 * (16 * jt + col));
                if (ti.x >= 0) dst[ti.x] = __builtin_fmaf(wo.x, O, we.x * E);
                if (ti.y >= 0) dst[ti.y] = __builtin_fmaf(wo.y, O, we.y * E);
                if (ti.z >= 0) dst[ti.z] = __builtin_fmaf(wo.z, O, we.z * E);
                if (ti.w >= 0) dst[ti.w] = __builtin_fmaf(wo.w, O, we.w * E);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {   // the tile's impulse responses are contiguous in ir (16 * Lw floats, 16-byte aligned start)
            float* dst = ir + (size_t)f0 * Lw;
            const int nfl = nf * Lw, n4 = nfl / 4;
            for (int i = lane; i < n4; i += 64)
                reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(reg)[i];
            for (int i = 4 * n4 + lane; i < nfl; i += 64) dst[i] = reg[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                 // rows copied out before the next tile's magnitudes land
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

__global__ void __launch_bounds__(256) fir_from_magnitudes_generic_kernel(const float* __restrict__ mags,
                                                                        const float* __restrict__ M,
                                                                        float* __restrict__ ir,
                                                                        size_t frames, int K, int Lw) {
    const size_t total = frames * Lw;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t f = g / Lw;
        const int tap = (int)(g - f * Lw);
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc = __builtin_fmaf(mags[f * K + k], M[(size_t)k * Lw + tap], acc);
        ir[g] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// time-varying FIR, tiled
// ------------------------------------------------------------------------------------------------
constexpr int FIR_OPL = 16;          // consecutive outputs per lane
constexpr int FIR_PASS = 16 * FIR_OPL;  // outputs per wavefront pass: 16 output groups x 4 input segments = 64 lanes
constexpr int FIR_G_FLOATS = 6400;   // per workgroup: staged frame FIRs (zero padded images)
constexpr int FIR_X_FLOATS = 2880;   // per workgroup: staged noise window (one pad block after every four)
// 37 KB of LDS per workgroup: four workgroups (16 wavefronts) per CU.

// LDS float offset of noise block bb (blocks of 4 samples, counted from the first staged one).  A 16-lane
// ds_read_b128 service group reads the blocks B, B + 4, ..., B + 60: with one pad block after every four
// they land 5 blocks apart, i.e. on 16 different bank quads (5 is coprime with 16) -- conflict free.
__device__ __forceinline__ int fir_xoff(int bb) { return 4 * (bb + (bb >> 2)); }

__global__ void __launch_bounds__(256) tv_fir_kernel(const float* __restrict__ x,   // [R, N]
                                                   const float* __restrict__ ir,  // [R, T, Lw]
                                                   float* __restrict__ out,       // [R, N]
                                                   int R, int N, int T, int U, int Lw, int delay,
                                                   int windows_per_row, int bw, int padl, int nb, int seglen,
                                                   int dc) {
    // One workgroup = bw consecutive outputs of one row: the frame FIRs and the noise samples that reach
    // them are staged ONCE for the four wavefronts (each then owns bw / 4 outputs).
    __shared__ __attribute__((aligned(16))) float lds[FIR_G_FLOATS + FIR_X_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int row = blockIdx.x / windows_per_row;
    const int nB0 = (blockIdx.x - row * windows_per_row) * bw;
    const int wlen = bw / 4;                      // outputs per wavefront
    const int n0 = nB0 + wib * wlen;
    float* G = lds;
    float* Xs = G + FIR_G_FLOATS;
    const int gstride = nb * 4;                   // floats per staged frame (<= 256)

    // frames whose noise blocks can reach this workgroup's outputs
    const int j_first = max(nB0 + delay - (Lw - 1), 0);
    const int j_last = min(nB0 + bw - 1 + delay, N - 1);
    const int f_lo = j_first / U;
    const int f_hi = min(j_last / U, T - 1);
    const int nfr = f_hi - f_lo + 1;
    // input blocks (of 4 samples) any lane may touch: zero outside the signal, so the inner loop needs no
    // bounds logic at all
    const int jb_base = (nB0 + delay + 3) / 4;
    const int jb_min = jb_base + 4 - 4 * seglen;                     // may be negative
    const int nxb = bw / 4 + 4 * seglen - 4;                         // staged blocks: jb_min .. jb_base + bw / 4 - 1
    // Staging: every load is issued before the first LDS store (branch-free clamped addresses), so a
    // wavefront pays the HBM/L2 latency once per batch of loads instead of once per load.
    {
        constexpr int XL = 3;                                       // 16-byte blocks per thread (nxb <= 768)
        const float4* xg = reinterpret_cast<const float4*>(x + (size_t)row * N);
        const int nblk = N / 4;
        float4 xv[XL];
#pragma unroll
        for (int u = 0; u < XL; ++u) xv[u] = xg[min(max(jb_min + (int)threadIdx.x + 256 * u, 0), nblk - 1)];
#pragma unroll
        for (int u = 0; u < XL; ++u) {
            const int bb = threadIdx.x + 256 * u, jb = jb_min + bb;
            if (bb < nxb)
                *reinterpret_cast<float4*>(Xs + fir_xoff(bb)) =
                    (jb >= 0 && jb < nblk) ? xv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    {
        constexpr int FB = 4;                       // frames per batch (per wavefront)
        constexpr int QL = 4;                       // 64-lane strips per staged frame (gstride <= 256)
        for (int f0 = wib * FB; f0 < nfr; f0 += 4 * FB) {
            float gv[FB][QL];
#pragma unroll
            for (int ff = 0; ff < FB; ++ff) {
                const float* src = ir + ((size_t)row * T + f_lo + min(f0 + ff, nfr - 1)) * Lw;
#pragma unroll
                for (int u = 0; u < QL; ++u) {
                    const int tap = lane + 64 * u - padl;
                    gv[ff][u] = src[min(max(tap, 0), Lw - 1)];
                }
            }
#pragma unroll
            for (int ff = 0; ff < FB; ++ff) {
#pragma unroll
                for (int u = 0; u < QL; ++u) {
                    const int q = lane + 64 * u, tap = q - padl;
                    if (f0 + ff < nfr && q < gstride)
                        G[(f0 + ff) * gstride + q] = (tap >= 0 && tap < Lw) ? gv[ff][u] : 0.0f;
                }
            }
        }
    }
    __syncthreads();

    // ---- compute: 256 outputs per pass.  Lane = (output group a, input segment sg): lane (a, sg) owns the 16
    // consecutive outputs of group a and the sg-th quarter of the ~(Lw + 18) / 4 input blocks that reach them.
    // Per step it takes one input block (4 samples) and the 19 taps that connect it to its 16 outputs: 64 FMAs
    // for six 16-byte LDS reads.  The lane -> (sg, a) map follows the way the LDS services a ds_read_b128 (four
    // fixed groups of 16 lanes): a group is one segment, so its tap reads are one address (broadcast) and its 16
    // noise blocks are 4 apart (conflict free through fir_xoff).
    const int bpf = U / 4;                               // input blocks per frame
    const int h = lane >> 5, w = lane & 31;
    const bool g0 = (w < 4) || (w >= 12 && w < 16) || (w >= 20 && w < 28);
    const int sg = 2 * h + (g0 ? 0 : 1);
    const int a = g0 ? (w < 4 ? w : (w < 16 ? w - 8 : w - 12)) : (w < 12 ? w - 4 : (w < 20 ? w - 8 : w - 16));
    const int bq0 = 4 * (dc - 3 + sg * seglen);          // float offset of the first tap block at step 0
    for (int np0 = n0; np0 < min(n0 + wlen, N); np0 += FIR_PASS) {
        const int m0 = np0 + delay;
        const int jb = (m0 + 3) / 4 + 4 * a + 3 - sg * seglen;   // first (highest) input block of this lane
        // A lane's seglen (<= U / 4) input blocks lie in at most two frames: the first n1 steps use the
        // frame of jb, the rest the frame before it.  Both tap pointers are per-lane constants, the
        // loop body is branch free (one select per step) and carries no state between steps.
        const int jbc = min(max(jb, 0), (N - 1) / 4);
        const int fA = min(max((4 * jbc) / U, f_lo), f_hi);
        const int n1 = jb - fA * bpf + 1;                // blocks past the end of the signal count as its last frame
        const float* GA = G + (fA - f_lo) * gstride + bq0;
        const float* GB = G + (max(fA - 1, f_lo) - f_lo) * gstride + bq0;
        const int bb0 = jb - jb_min;
        float acc[FIR_OPL];
#pragma unroll
        for (int e = 0; e < FIR_OPL; ++e) acc[e] = 0.f;
        // software pipelined: the six LDS reads of step i + 1 are in flight while the 64 FMAs of step i issue
        float4 tq[5], xq;
        auto fetch = [&](int i) {
            const float* gp = (i < n1 ? GA : GB) + 4 * i;
#pragma unroll
            for (int q = 0; q < 5; ++q) tq[q] = *reinterpret_cast<const float4*>(gp + 4 * q);
            xq = *reinterpret_cast<const float4*>(Xs + fir_xoff(bb0 - i));
        };
        fetch(0);
#pragma unroll 1
        for (int i = 0; i < seglen; ++i) {
            float tp[20];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                tp[4 * q] = tq[q].x; tp[4 * q + 1] = tq[q].y; tp[4 * q + 2] = tq[q].z; tp[4 * q + 3] = tq[q].w;
            }
            asm volatile("" ::"v"(tp[19]));   // keep the fifth read 16 bytes wide: ds_read_b96 costs 8 LDS cycles, b128 4
            const float xs[4] = {xq.x, xq.y, xq.z, xq.w};
            fetch(min(i + 1, seglen - 1));
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int e = 0; e < FIR_OPL; ++e) acc[e] = __builtin_fmaf(xs[d], tp[e - d + 3], acc[e]);
        }
        // meet the four segments of each output group (reduce-scatter in registers): the lane pairs (l, l ^ 32)
        // and (l, l ^ 4) hold the same group a; each lane ends up with the 4 outputs 4 sg .. 4 sg + 3 of the group
        float half[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float keep = h ? acc[8 + k] : acc[k];
            const float send = h ? acc[k] : acc[8 + k];
            half[k] = keep + __shfl_xor(send, 32);
        }
        float quad[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float keep = g0 ? half[k] : half[4 + k];
            const float send = g0 ? half[4 + k] : half[k];
            quad[k] = keep + __shfl_xor(send, 4);
        }
        const int n = np0 + FIR_OPL * a + 4 * sg;
        if (n < N) *reinterpret_cast<float4*>(out + (size_t)row * N + n) = make_float4(quad[0], quad[1], quad[2], quad[3]);
    }
}

// ------------------------------------------------------------------------------------------------
// FilteredNoise in one kernel: FIR design (matrix cores) + time-varying FIR, impulse responses never leave the CU
// ------------------------------------------------------------------------------------------------
// Persistent workgroups walk (row, window of 1024 outputs) tasks.  Per window:
//   1. the noise blocks and the magnitudes of the (<= 16) frames that reach the window, prefetched into registers
//      during the previous window's arithmetic, go to LDS (scale_fn applied to raw magnitudes on the way);
//   2. E = M_even CE, O = M_odd CO on v_mfma_f32_16x16x4_f32: wavefront jt < JT owns the 16-column block jt, its
//      table fragments stay in registers for the life of the workgroup;
//   3. tap weights straight from the accumulators: every (frame, column) pair yields four taps of the frame's
//      zero-padded FIR image in LDS (the zero pads are written once per workgroup);
//   4. the time-varying FIR of tv_fir_kernel: each wavefront 256 outputs, 64 FMAs per six ds_read_b128.
// HBM sees the noise, 1.4 x the magnitudes and the output: the [R, T, Lw] impulse responses (0.58 GB written
// and 0.8 GB read at batch 64) do not exist.
constexpr int FUS_BW = 1024;            // outputs per window
constexpr int FUS_FRAMES = 16;          // frames staged per window = one MFMA row tile

template <int KH, int JT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
noise_fir_fused_kernel(const float* __restrict__ x,          // [R, N] noise
                       const float* __restrict__ mags,       // [R, T, 2 KH]
                       const float* __restrict__ CE, const float* __restrict__ CO,     // [KH, NJ]
                       const int* __restrict__ tap_idx, const float* __restrict__ tap_we,
                       const float* __restrict__ tap_wo,     // [NJ, 4]
                       float* __restrict__ out,              // [R / vq, N]
                       float* __restrict__ out_last,         // [R / n_voices, N] the last voice of every segment on its own (vq > 1), or null
                       int R, int N, int T, int U, int Lw, int NJ, int delay, int windows_per_row, int padl,
                       int nb, int seglen, int dc, float bias, ScaleFn scale, int vq, int n_voices, int vmajor) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int K = 2 * KH, KS = KH / 4, ASTR = K + 4, NJP = 16 * JT;
    extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
    const int gstride = nb * 4;                               // floats per staged frame image (<= 256)
    float* G = lds_dyn;                                       // [16][gstride]
    float* Xs = G + FUS_FRAMES * 256;                         // padded noise window
    float* M = Xs + 1536;                                     // [16][ASTR] magnitudes
    int* tix = reinterpret_cast<int*>(M + FUS_FRAMES * ASTR);       // [NJP][4]
    float* twe = reinterpret_cast<float*>(tix + 4 * NJP);
    float* two = twe + 4 * NJP;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;

    // ---- once per workgroup: tap tables, zero pads of the FIR images, this wavefront's table fragments
    for (int i = threadIdx.x; i < 4 * NJP; i += 256) {
        const int j = i >> 2;
        tix[i] = j < NJ ? tap_idx[i] : -1;
        twe[i] = j < NJ ? tap_we[i] : 0.0f;
        two[i] = j < NJ ? tap_wo[i] : 0.0f;
    }
    for (int i = threadIdx.x; i < FUS_FRAMES * 256; i += 256) G[i] = 0.0f;
    // wavefront jt < JT owns the 16-column block jt: E and O fragments of the tables, and that block's tap weights
    const int jt = min(wib, JT - 1);
    float bE[KS], bO[KS];
    {
        const int j = min(16 * jt + col, NJ - 1);
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            bE[st] = CE[(4 * st + kq) * NJ + j];
            bO[st] = CO[(4 * st + kq) * NJ + j];
        }
    }

    const int bpf = U / 4;                                    // input blocks per frame
    const int nblk = N / 4;
    const int nxb = FUS_BW / 4 + 4 * seglen - 4;              // staged noise blocks per window (<= 512)
    const int h = lane >> 5, w = lane & 31;
    const bool g0 = (w < 4) || (w >= 12 && w < 16) || (w >= 20 && w < 28);
    const int sg = 2 * h + (g0 ? 0 : 1);
    const int a = g0 ? (w < 4 ? w : (w < 16 ? w - 8 : w - 12)) : (w < 12 ? w - 4 : (w < 20 ? w - 8 : w - 16));
    const int bq0 = 4 * (dc - 3 + sg * seglen);
    // vq > 1: the filtered noise of vq consecutive voices of a segment is summed in registers and leaves as ONE row
    // (out[b, q] = sum_i voice q vq + i), a quarter of the writes and of the mixer's reads for vq = 4.  The vq
    // (voice, window) units of an output window are walked back to back by the same workgroup.
    const int ntasks = (R / vq) * windows_per_row;             // output (row, window) tasks
    const int n_seg = R / n_voices, pq = n_voices / vq;        // segments, output rows per segment

    // registers holding the NEXT window's inputs
    float4 xv[2], mv[2];
    auto window_geometry = [&](int task, int iv, int& row, int& nB0, int& f_lo, int& jb_min) {
        const int orow = task / windows_per_row;
        if (vq == 1) {
            row = orow;
        } else {
            const int b = orow / pq, v = (orow - b * pq) * vq + iv;
            row = vmajor ? v * n_seg + b : b * n_voices + v;
        }
        nB0 = (task - orow * windows_per_row) * FUS_BW;
        f_lo = max(nB0 + delay - (Lw - 1), 0) / U;
        jb_min = (nB0 + delay + 3) / 4 + 4 - 4 * seglen;
    };
    auto prefetch = [&](int task, int iv) {
        int row, nB0, f_lo, jb_min;
        window_geometry(task, iv, row, nB0, f_lo, jb_min);
        const float4* xg = reinterpret_cast<const float4*>(x + (size_t)row * N);
#pragma unroll
        for (int u = 0; u < 2; ++u) xv[u] = xg[min(max(jb_min + (int)threadIdx.x + 256 * u, 0), nblk - 1)];
        constexpr int PER_ROW = K / 4;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = min((int)threadIdx.x + 256 * u, FUS_FRAMES * PER_ROW - 1);
            const int fr = i / PER_ROW, c4 = i - fr * PER_ROW;
            const int f = min(f_lo + fr, T - 1);
            mv[u] = reinterpret_cast<const float4*>(mags + ((size_t)row * T + f) * K)[c4];
        }
    };
    if ((int)blockIdx.x < ntasks) prefetch(blockIdx.x, 0);
    __syncthreads();
    float4 vsum = make_float4(0.f, 0.f, 0.f, 0.f);            // this lane's four outputs, summed over the vq voices
    for (int task = blockIdx.x; task < ntasks; task += gridDim.x)
    for (int iv = 0; iv < vq; ++iv) {
        int row, nB0, f_lo, jb_min;
        window_geometry(task, iv, row, nB0, f_lo, jb_min);
        const int f_hi = min(min(nB0 + FUS_BW - 1 + delay, N - 1) / U, T - 1);
        // ---- 1. registers -> LDS
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int bb = threadIdx.x + 256 * u, jb = jb_min + bb;
            if (bb < nxb)
                *reinterpret_cast<float4*>(Xs + fir_xoff(bb)) =
                    (jb >= 0 && jb < nblk) ? xv[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        {
            constexpr int PER_ROW = K / 4;
            with_scale_kind(scale.kind, [&](auto kind) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = threadIdx.x + 256 * u;
                    if (i < FUS_FRAMES * PER_ROW) {
                        const float4 m = scale4_of<decltype(kind)::value>(scale, mv[u], bias);
                        const int fr = i / PER_ROW, c4 = i - fr * PER_ROW;
                        *reinterpret_cast<float4*>(M + fr * ASTR + 4 * c4) = m;
                    }
                }
            });
        }
        __syncthreads();
        if (iv + 1 < vq) prefetch(task, iv + 1);                              // in flight during steps 2-4
        else if (task + (int)gridDim.x < ntasks) prefetch(task + gridDim.x, 0);
        // ---- 2. E / O blocks on the matrix cores, 3. tap weights -> zero-padded FIR images (straight from the
        // accumulators: lane holds E, O of frames 4 kq .. 4 kq + 3 at column 16 jt + col)
        if (wib < JT) {
            f32x4 accE = f32x4{0.f, 0.f, 0.f, 0.f}, accO = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* arow = M + col * ASTR + 2 * kq;
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                const float2 am = *reinterpret_cast<const float2*>(arow + 8 * st);
                accE = __builtin_amdgcn_mfma_f32_16x16x4f32(am.x, bE[st], accE, 0, 0, 0);
                accO = __builtin_amdgcn_mfma_f32_16x16x4f32(am.y, bO[st], accO, 0, 0, 0);
            }
            const int j = 16 * jt + col;
            const int4 ti = *reinterpret_cast<const int4*>(tix + 4 * j);
            const float4 we = *reinterpret_cast<const float4*>(twe + 4 * j);
            const float4 wo = *reinterpret_cast<const float4*>(two + 4 * j);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float E = accE[r], O = accO[r];
                float* dst = G + (4 * kq + r) * gstride + padl;
                if (ti.x >= 0) dst[ti.x] = __builtin_fmaf(wo.x, O, we.x * E);
                if (ti.y >= 0) dst[ti.y] = __builtin_fmaf(wo.y, O, we.y * E);
                if (ti.z >= 0) dst[ti.z] = __builtin_fmaf(wo.z, O, we.z * E);
                if (ti.w >= 0) dst[ti.w] = __builtin_fmaf(wo.w, O, we.w * E);
            }
        }
        __syncthreads();
        // ---- 4. time-varying FIR: this wavefront's 256 outputs (see tv_fir_kernel)
        const int np0 = nB0 + wib * FIR_PASS;
        if (np0 < N) {
            const int m0 = np0 + delay;
            const int jb = (m0 + 3) / 4 + 4 * a + 3 - sg * seglen;
            const int jbc = min(max(jb, 0), (N - 1) / 4);
            const int fA = min(max((4 * jbc) / U, f_lo), f_hi);
            const int n1 = jb - fA * bpf + 1;
            const float* GA = G + (fA - f_lo) * gstride + bq0;
            const float* GB = G + (max(fA - 1, f_lo) - f_lo) * gstride + bq0;
            const int bb0 = jb - jb_min;
            float acc[FIR_OPL];
#pragma unroll
            for (int e = 0; e < FIR_OPL; ++e) acc[e] = 0.f;
            float4 tq[5], xq;
            auto fetch = [&](int i) {
                const float* gp = (i < n1 ? GA : GB) + 4 * i;
#pragma unroll
                for (int q = 0; q < 5; ++q) tq[q] = *reinterpret_cast<const float4*>(gp + 4 * q);
                xq = *reinterpret_cast<const float4*>(Xs + fir_xoff(bb0 - i));
            };
            fetch(0);
#pragma unroll 1
            for (int i = 0; i < seglen; ++i) {
                float tp[20];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    tp[4 * q] = tq[q].x; tp[4 * q + 1] = tq[q].y; tp[4 * q + 2] = tq[q].z; tp[4 * q + 3] = tq[q].w;
                }
                asm volatile("" ::"v"(tp[19]));
                const float xs[4] = {xq.x, xq.y, xq.z, xq.w};
                fetch(min(i + 1, seglen - 1));
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int e = 0; e < FIR_OPL; ++e) acc[e] = __builtin_fmaf(xs[d], tp[e - d + 3], acc[e]);
            }
            float half[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float keep = h ? acc[8 + k] : acc[k];
                const float send = h ? acc[k] : acc[8 + k];
                half[k] = keep + __shfl_xor(send, 32);
            }
            float quad[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float keep = g0 ? half[k] : half[4 + k];
                const float send = g0 ? half[4 + k] : half[k];
                quad[k] = keep + __shfl_xor(send, 4);
            }
            const int n = np0 + FIR_OPL * a + 4 * sg;
            // out_last: the segment's last voice (the last unit of its last output row) leaves on its own and stays out
            // of the sum -- the outputs dictionary of the reference's DAG holds that voice's noise next to the mix
            const int orow = task / windows_per_row;
            const bool lastv = out_last != nullptr && iv == vq - 1 && (orow % pq) == pq - 1;
            if (iv == 0) vsum = make_float4(quad[0], quad[1], quad[2], quad[3]);
            else if (!lastv) { vsum.x += quad[0]; vsum.y += quad[1]; vsum.z += quad[2]; vsum.w += quad[3]; }
            if (iv == vq - 1 && n < N) {
                *reinterpret_cast<float4*>(out + (size_t)orow * N + n) = vsum;
                if (lastv) *reinterpret_cast<float4*>(out_last + (size_t)(orow / pq) * N + n) = make_float4(quad[0], quad[1], quad[2], quad[3]);
            }
        }
        __syncthreads();                 // LDS is rewritten by the next window
    }
}

__global__ void __launch_bounds__(256) tv_fir_generic_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ ir,
                                                           float* __restrict__ out, int R, int N, int T,
                                                           int U, int Lw, int delay) {
    const size_t total = (size_t)R * N;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int row = (int)(g / N), n = (int)(g - (size_t)row * N);
        const int m = n + delay;
        const int j_lo = max(m - (Lw - 1), 0), j_hi = min(m, N - 1);
        float acc = 0.0f;
        for (int j = j_lo; j <= j_hi; ++j) {
            const int f = min(j / U, T - 1);
            acc = __builtin_fmaf(x[(size_t)row * N + j], ir[((size_t)row * T + f) * Lw + (m - j)], acc);
        }
        out[g] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 uniform noise in [-1, 1)
// ------------------------------------------------------------------------------------------------
// (philox_round / philox_uniform4: noise_win.h)
__global__ void __launch_bounds__(256) uniform_noise_kernel(float* __restrict__ out, size_t n4,
                                                          uint64_t seed, uint64_t offset) {
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < n4; g += (size_t)gridDim.x * 256) {
        reinterpret_cast<float4*>(out)[g] = philox_uniform4(seed, offset + g);
    }
}

// rows of a [R, n] tensor, row r drawn from counters offset + r * row_stride + i / 4: a (row, position)-keyed stream
__global__ void __launch_bounds__(256) uniform_noise_rows_kernel(float* __restrict__ out, int R, size_t n4, uint64_t seed,
                                                               uint64_t offset, uint64_t row_stride) {
    const size_t total = (size_t)R * n4;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t r = g / n4, i = g - r * n4;
        reinterpret_cast<float4*>(out)[g] = philox_uniform4(seed, offset + r * row_stride + i);
    }
}

static unsigned stream_grid(size_t total) {
    size_t blocks = (total + 255) / 256;
    const size_t cap = 256 * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

static int env_int(const char* name, int dflt) { return ddspp_option_literal(name, dflt); }

}  // namespace ddspp

using namespace ddspp;

extern "C" {

static bool win_tvfir_shape(int N, int T, int Lw, int delay, WinGeom* g);

// ddsp.core.frequency_impulse_response(magnitudes, window_size) as one product with the host-built
// matrix M[K, Lw] (ddsp_piano_amd/core.py: irfft basis x window, shifted to causal form).
// uniq[n_uniq] / mirror[n_uniq] (device int32, may be NULL): the taps to compute and where each
// one is mirrored to (-1: nowhere); NULL means every tap is computed.
int ddspp_fir_from_magnitudes(const float* magnitudes, const float* M, const int* uniq, const int* mirror,
                              int n_uniq, float* ir, size_t frames, int K, int Lw, hipStream_t stream) {
    DDSPP_REQUIRE(magnitudes && M && ir, "fir_from_magnitudes: null buffer");
    DDSPP_REQUIRE(K > 0 && Lw > 0, "fir_from_magnitudes: bad dims");
    DDSPP_REQUIRE(frames < (1ull << 31), "fir_from_magnitudes: too many frames");
    if (frames == 0) return DDSPP_OK;
    const bool tiled = uniq && mirror && n_uniq > 0 && n_uniq <= 256 &&
                       (K == 32 || K == 64 || K == 96 || K == 128) && !env_int("DDSPP_FIR_GENERIC", 0);
    if (tiled) {
        const int fpb = env_int("DDSPP_FIR_FPB", 256);
        const dim3 grid((unsigned)((frames + fpb - 1) / fpb));
        const int nf = (int)frames;
#define DDSPP_FIR_LAUNCH(KK)                                                                              \
        do {                                                                                              \
            if (n_uniq <= 128)                                                                            \
                hipLaunchKernelGGL((fir_from_magnitudes_kernel<KK, 128>), grid, dim3(128), 0, stream,     \
                                   magnitudes, M, uniq, mirror, ir, nf, Lw, n_uniq, fpb);                 \
            else                                                                                          \
                hipLaunchKernelGGL((fir_from_magnitudes_kernel<KK, 256>), grid, dim3(256), 0, stream,     \
                                   magnitudes, M, uniq, mirror, ir, nf, Lw, n_uniq, fpb);                 \
        } while (0)
        if (K == 32) DDSPP_FIR_LAUNCH(32);
        else if (K == 64) DDSPP_FIR_LAUNCH(64);
        else if (K == 96) DDSPP_FIR_LAUNCH(96);
        else DDSPP_FIR_LAUNCH(128);
#undef DDSPP_FIR_LAUNCH
    } else {
        hipLaunchKernelGGL(fir_from_magnitudes_generic_kernel, dim3(stream_grid(frames * Lw)), dim3(256), 0,
                           stream, magnitudes, M, ir, frames, K, Lw);
    }
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// ddsp.core.fft_convolve(audio, impulse_response[B, T, Lw], padding='same', delay_compensation)
// for the framed (time-varying) case: frame_size = hop = U = N / T.
int ddspp_time_varying_fir(const float* audio, const float* impulse_response, float* out, int R, int N,
                           int T, int Lw, int delay_compensation, hipStream_t stream) {
    DDSPP_REQUIRE(audio && impulse_response && out, "time_varying_fir: null buffer");
    DDSPP_REQUIRE(R > 0 && N > 0 && T > 0 && Lw > 0, "time_varying_fir: bad dims");
    DDSPP_REQUIRE(N % T == 0, "time_varying_fir: n_samples=%d must be a multiple of n_frames=%d", N, T);
    const int U = N / T;
    const int delay = ddspp_auto_delay(delay_compensation, Lw);
    DDSPP_REQUIRE(delay >= 0, "time_varying_fir: negative delay");
    {
        WinGeom wg;
        if ((uintptr_t)audio % 16 == 0 && (uintptr_t)out % 16 == 0 && !env_int("DDSPP_FIR_GENERIC", 0) &&
            win_tvfir_shape(N, T, Lw, delay, &wg))
            return launch_win_tvfir(audio, impulse_response, out, R, N, T, Lw, wg, stream);
    }
    // Tiled kernel geometry (see tv_fir_kernel).  Tap t of a frame sits at float padl + t of its staged image;
    // padl makes (delay - 3 + padl) a multiple of 4 (aligned 16-byte tap blocks) and puts enough zero blocks in
    // front that segment 0 starts at block dc - 3 >= 0; nb leaves zero blocks behind for every block index the
    // inner loop can form (no clamping in the loop).
    int padl = 18 - (delay + 3) % 4;
    while ((delay - 3 + padl) % 4 != 0) ++padl;
    const int dc = (padl - 6 + (delay + 3) % 4) / 4;
    const int seglen = (((Lw + FIR_OPL - 1 + 3 + 3) / 4) + 3) / 4;       // steps per lane: a quarter of the input blocks
    const int nb_need = (padl + Lw + 3) / 4 + 1;
    const int nb = (dc + 4 * seglen + 2 > nb_need ? dc + 4 * seglen + 2 : nb_need);
    // outputs per workgroup: 2048, or 1024 when the frames are short (more of them per window)
    int bw = 0;
    for (int cand : {2048, 1024}) {
        const int frames_max = (cand + Lw + U - 2) / U + 2;
        const int nxb = cand / 4 + 4 * seglen - 4;
        if (frames_max * nb * 4 <= FIR_G_FLOATS && (nxb + nxb / 4 + 2) * 4 <= FIR_X_FLOATS && nxb <= 768) {
            bw = cand;
            break;
        }
    }
    const bool tiled = bw > 0 && (U % 4 == 0) && (N % 4 == 0) && ((uintptr_t)audio % 16 == 0) &&
                       ((uintptr_t)out % 16 == 0) && dc >= 3 && nb * 4 <= 256 && seglen <= U / 4 &&
                       !env_int("DDSPP_FIR_GENERIC", 0);
    if (tiled) {
        const int wpr = (N + bw - 1) / bw;
        const long long tasks = (long long)R * wpr;
        DDSPP_REQUIRE(tasks < (1ll << 31), "time_varying_fir: too many tasks");
        hipLaunchKernelGGL(tv_fir_kernel, dim3((unsigned)tasks), dim3(256), 0, stream, audio,
                           impulse_response, out, R, N, T, U, Lw, delay, wpr, bw, padl, nb, seglen, dc);
    } else {
        hipLaunchKernelGGL(tv_fir_generic_kernel, dim3(stream_grid((size_t)R * N)), dim3(256), 0, stream,
                           audio, impulse_response, out, R, N, T, U, Lw, delay);
    }
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// Same operator through the even/odd tables (host: ddsp_piano_amd/core.py::_fir_eo_tables);
// K must be even and a multiple of 4, NJ <= 64.  scale_kind >= 0: `magnitudes` are the raw network outputs and
// FilteredNoise.get_controls' scale_fn(magnitudes + bias) is applied on the way in (saves one HBM round trip
// of the [frames, K] tensor); scale_kind = -1: magnitudes are used as given.
int ddspp_fir_from_magnitudes_eo(const float* magnitudes, const float* CE, const float* CO, const int* tap_idx,
                                 const float* tap_we, const float* tap_wo, float* ir, size_t frames, int K,
                                 int Lw, int NJ, int scale_kind, float bias, float exponent, float max_value,
                                 float threshold, float gain, hipStream_t stream) {
    DDSPP_REQUIRE(magnitudes && CE && CO && tap_idx && tap_we && tap_wo && ir, "fir_from_magnitudes_eo: null buffer");
    DDSPP_REQUIRE(scale_kind >= -1 && scale_kind <= 2, "fir_from_magnitudes_eo: unknown scale_fn %d", scale_kind);
    DDSPP_REQUIRE(K > 0 && K % 4 == 0 && NJ > 0 && NJ <= 64 && Lw > 0, "fir_from_magnitudes_eo: bad dims");
    DDSPP_REQUIRE(K == 32 || K == 64 || K == 96 || K == 128, "fir_from_magnitudes_eo: unsupported band count %d", K);
    DDSPP_REQUIRE(frames < (1ull << 31), "fir_from_magnitudes_eo: too many frames");
    DDSPP_REQUIRE((uintptr_t)magnitudes % 16 == 0, "fir_from_magnitudes_eo: magnitudes must be 16-byte aligned");
    if (frames == 0) return DDSPP_OK;
    if (K <= 96 && NJ <= 16 * ((K / 2 + 15) / 16) && !env_int("DDSPP_FIR_NO_MFMA", 0)) {
        DDSPP_REQUIRE((uintptr_t)ir % 16 == 0, "fir_from_magnitudes_eo: ir must be 16-byte aligned");
        const int region = ((16 * (K + 4) > 16 * Lw ? 16 * (K + 4) : 16 * Lw) + 3) / 4 * 4;
        const size_t lds_m = ((size_t)4 * region + 3 * 64 * 4) * sizeof(float);       // + tap tables
        const long long ntiles16 = ((long long)frames + 15) / 16;
        long long wgs = (ntiles16 + 3) / 4;
        const long long cap = (long long)256 * env_int("DDSPP_FIR_WGS_PER_CU", 3);    // wavefronts walk tiles beyond that
        if (wgs > cap) wgs = cap;
        const dim3 grid_m((unsigned)wgs), block_m(256);
        const ScaleFn sf{scale_kind, scale_kind > 0 ? logf(exponent) : 0.0f, max_value, threshold, gain};
        const int nfr = (int)frames;
        if (K == 32) hipLaunchKernelGGL((fir_eo_mfma_kernel<16, 1>), grid_m, block_m, lds_m, stream, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, ir, nfr, Lw, NJ, region, bias, sf);
        else if (K == 64) hipLaunchKernelGGL((fir_eo_mfma_kernel<32, 2>), grid_m, block_m, lds_m, stream, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, ir, nfr, Lw, NJ, region, bias, sf);
        else hipLaunchKernelGGL((fir_eo_mfma_kernel<48, 3>), grid_m, block_m, lds_m, stream, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, ir, nfr, Lw, NJ, region, bias, sf);
        DDSPP_LAUNCH_CHECK();
        return DDSPP_OK;
    }
    const int fpb = 32;               // frames per tile: (K + Lw) * 4 * 32 bytes of LDS, four workgroups per CU
    DDSPP_REQUIRE(((size_t)fpb * Lw) % 4 == 0 && (uintptr_t)ir % 16 == 0, "fir_from_magnitudes_eo: ir must be 16-byte aligned");
    const long long ntiles = ((long long)frames + fpb - 1) / fpb;
    const int tpw = env_int("DDSPP_FIR_TILES_PER_WG", 4);       // tiles a workgroup walks
    const dim3 grid((unsigned)((ntiles + tpw - 1) / (tpw > 0 ? tpw : 1))), block(256);
    const size_t lds = (size_t)fpb * (K + Lw) * sizeof(float);
    const int nf = (int)frames;
    const ScaleFn sfn{scale_kind, scale_kind > 0 ? logf(exponent) : 0.0f, max_value, threshold, gain};
    if (K == 32) hipLaunchKernelGGL(fir_eo_kernel<16>, grid, block, lds, stream, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, ir, nf, Lw, NJ, fpb, bias, sfn);
    else if (K == 64) hipLaunchKernelGGL(fir_eo_kernel<32>, grid, block, lds, stream, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, ir, nf, Lw, NJ, fpb, bias, sfn);
    else if (K == 96) hipLaunchKernelGGL(fir_eo_kernel<48>, grid, block, lds, stream, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, ir, nf, Lw, NJ, fpb, bias, sfn);
    else hipLaunchKernelGGL(fir_eo_kernel<64>, grid, block, lds, stream, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, ir, nf, Lw, NJ, fpb, bias, sfn);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// Geometry of the fused FilteredNoise kernel for a shape, or false when it does not apply (the caller then
// runs ddspp_fir_from_magnitudes_eo + ddspp_time_varying_fir).
struct FusedGeom {
    int U, delay, padl, dc, seglen, nb, wpr;
};
static bool fused_shape(int N, int T, int K, int Lw, int delay_compensation, FusedGeom* g) {
    if (N <= 0 || T <= 0 || N % T != 0 || (K != 32 && K != 64 && K != 96) || Lw != 2 * (K - 1)) return false;
    const int U = N / T;
    const int delay = ddspp_auto_delay(delay_compensation, Lw);
    if (delay < 0 || U % 4 != 0 || N % 4 != 0) return false;
    int padl = 18 - (delay + 3) % 4;
    while ((delay - 3 + padl) % 4 != 0) ++padl;
    const int dc = (padl - 6 + (delay + 3) % 4) / 4;
    const int seglen = (((Lw + FIR_OPL - 1 + 3 + 3) / 4) + 3) / 4;
    const int nb_need = (padl + Lw + 3) / 4 + 1;
    const int nb = (dc + 4 * seglen + 2 > nb_need ? dc + 4 * seglen + 2 : nb_need);
    const int frames_max = (FUS_BW + Lw + U - 2) / U + 2;
    const int nxb = FUS_BW / 4 + 4 * seglen - 4;
    if (frames_max > FUS_FRAMES || nb * 4 > 256 || nxb > 512 || 4 * (nxb + nxb / 4) + 4 > 1536 || dc < 3 ||
        seglen > U / 4)
        return false;
    *g = FusedGeom{U, delay, padl, dc, seglen, nb, (N + FUS_BW - 1) / FUS_BW};
    return true;
}
static bool fused_geometry(int N, int T, int K, int Lw, int delay_compensation, FusedGeom* g) {
    return !env_int("DDSPP_FIR_NO_FUSED", 0) && fused_shape(N, T, K, Lw, delay_compensation, g);
}
// The windowed kernels (noise_win.hip) take a shape in both forms or in neither, so that the fused and the two-call
// form of a shape always run the same walk (bit-identical results).  DDSPP_FIR_WIN=0: round 2's kernels.
static bool win_fused_shape(int N, int T, int K, int Lw, int delay_compensation, WinGeom* g) {
    if (N <= 0 || T <= 0 || N % T != 0 || !env_int("DDSPP_FIR_WIN", 1)) return false;
    return win_fused_supported(N, T, K, Lw, ddspp_auto_delay(delay_compensation, Lw), g);
}
static bool win_tvfir_shape(int N, int T, int Lw, int delay, WinGeom* g) {
    if (!env_int("DDSPP_FIR_WIN", 1) || !win_tvfir_supported(N, T, Lw, delay, g)) return false;
    FusedGeom fg;
    WinGeom wg;
    const int K = Lw / 2 + 1;
    if (Lw == 2 * (K - 1) && fused_shape(N, T, K, Lw, delay, &fg) && !win_fused_supported(N, T, K, Lw, delay, &wg))
        return false;          // round 2's fused kernel takes this shape: its two-call form must match it
    return true;
}

int ddspp_frequency_filter_eo_supported(int N, int T, int K, int Lw, int delay_compensation) {
    FusedGeom g;
    WinGeom wg;
    if (env_int("DDSPP_FIR_NO_FUSED", 0)) return 0;
    return (win_fused_shape(N, T, K, Lw, delay_compensation, &wg) || fused_shape(N, T, K, Lw, delay_compensation, &g)) ? 1 : 0;
}

static int launch_fused_noise(const float* audio, const float* magnitudes, const float* CE, const float* CO,
                              const int* tap_idx, const float* tap_we, const float* tap_wo, float* out, float* out_last,
                              int R, int N,
                              int T, int K, int Lw, int NJ, int delay_compensation, int scale_kind, float bias,
                              float exponent, float max_value, float threshold, float gain, int vq, int n_voices,
                              int voice_major, hipStream_t stream, bool draw = false, uint64_t draw_seed = 0,
                              uint64_t draw_offset = 0) {
    DDSPP_REQUIRE((audio || draw) && magnitudes && CE && CO && tap_idx && tap_we && tap_wo && out,
                  "frequency_filter_eo: null buffer");
    DDSPP_REQUIRE(R > 0, "frequency_filter_eo: bad dims");
    DDSPP_REQUIRE(scale_kind >= -1 && scale_kind <= 2, "frequency_filter_eo: unknown scale_fn %d", scale_kind);
    DDSPP_REQUIRE(vq >= 1 && n_voices >= 1 && n_voices % vq == 0 && R % n_voices == 0,
                  "frequency_filter_eo: %d voices per output row do not divide %d voices / %d rows", vq, n_voices, R);
    DDSPP_REQUIRE(!out_last || (vq > 1 && (uintptr_t)out_last % 16 == 0),
                  "frequency_filter_eo: out_last needs voices_per_row > 1 and a 16-byte aligned buffer");
    DDSPP_REQUIRE((uintptr_t)audio % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)magnitudes % 16 == 0,
                  "frequency_filter_eo: buffers must be 16-byte aligned");
    WinGeom wg;
    if (NJ == K / 2 && !env_int("DDSPP_FIR_NO_FUSED", 0) && win_fused_shape(N, T, K, Lw, delay_compensation, &wg)) {
        const ScaleFn wsf{scale_kind, scale_kind > 0 ? logf(exponent) : 0.0f, max_value, threshold, gain};
        return launch_win_fused(audio, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, out, out_last, R, N, T, K, NJ, wg, bias,
                                wsf, vq, n_voices, voice_major, stream, draw, draw_seed, draw_offset);
    }
    DDSPP_REQUIRE(!draw, "frequency_filter_eo_voices_drawn: only the windowed kernel draws its own noise "
                  "(ddspp_frequency_filter_eo_drawn_supported says for which shapes): N=%d T=%d K=%d Lw=%d", N, T, K, Lw);
    FusedGeom g;
    DDSPP_REQUIRE(fused_geometry(N, T, K, Lw, delay_compensation, &g) && NJ == K / 2,
                  "frequency_filter_eo: shape not supported (N=%d T=%d K=%d Lw=%d)", N, T, K, Lw);
    const long long tasks = (long long)(R / vq) * g.wpr;
    DDSPP_REQUIRE((long long)R * g.wpr < (1ll << 31), "frequency_filter_eo: too many tasks");
    const int njp = 16 * ((NJ + 15) / 16);
    const size_t lds = ((size_t)FUS_FRAMES * 256 + 1536 + (size_t)FUS_FRAMES * (K + 4) + 3 * 4 * njp) * sizeof(float);
    long long wgs = (long long)256 * env_int("DDSPP_FUSED_WGS_PER_CU", 32);
    if (wgs > tasks) wgs = tasks;
    const dim3 grid((unsigned)wgs), block(256);
    const ScaleFn sf{scale_kind, scale_kind > 0 ? logf(exponent) : 0.0f, max_value, threshold, gain};
#define DDSPP_FUSED_LAUNCH(KH, JT)                                                                              \
    hipLaunchKernelGGL((noise_fir_fused_kernel<KH, JT>), grid, block, lds, stream, audio, magnitudes, CE, CO,    \
                       tap_idx, tap_we, tap_wo, out, out_last, R, N, T, g.U, Lw, NJ, g.delay, g.wpr, g.padl, g.nb, \
                       g.seglen, g.dc, bias, sf, vq, n_voices, voice_major)
    if (K == 32) DDSPP_FUSED_LAUNCH(16, 1);
    else if (K == 64) DDSPP_FUSED_LAUNCH(32, 2);
    else DDSPP_FUSED_LAUNCH(48, 3);
#undef DDSPP_FUSED_LAUNCH
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// ddsp.core.frequency_filter (frequency_impulse_response + framed fft_convolve) in one kernel; call only for
// shapes ddspp_frequency_filter_eo_supported accepts.  Tables and scale arguments as ddspp_fir_from_magnitudes_eo.
int ddspp_frequency_filter_eo(const float* audio, const float* magnitudes, const float* CE, const float* CO,
                              const int* tap_idx, const float* tap_we, const float* tap_wo, float* out, int R,
                              int N, int T, int K, int Lw, int NJ, int delay_compensation, int scale_kind,
                              float bias, float exponent, float max_value, float threshold, float gain,
                              hipStream_t stream) {
    return launch_fused_noise(audio, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, out, nullptr, R, N, T, K, Lw, NJ,
                              delay_compensation, scale_kind, bias, exponent, max_value, threshold, gain, 1, 1, 0,
                              stream);
}

// The same for the rows of a polyphonic group (rows = n_segments x n_voices, segment major or voice major), with the
// filtered noise of `voices_per_row` consecutive voices summed into one output row: out[R / voices_per_row, N],
// out[b, q] = sum_i filtered(voice q * voices_per_row + i of segment b).  Feeds ddspp_mix_voices.
// out_last (may be null) [R / n_voices, N]: the last voice of every segment goes there instead of into its row's sum.
int ddspp_frequency_filter_eo_voices(const float* audio, const float* magnitudes, const float* CE, const float* CO,
                                     const int* tap_idx, const float* tap_we, const float* tap_wo, float* out,
                                     float* out_last,
                                     int R, int N, int T, int K, int Lw, int NJ, int delay_compensation,
                                     int scale_kind, float bias, float exponent, float max_value, float threshold,
                                     float gain, int n_voices, int voices_per_row, int voice_major,
                                     hipStream_t stream) {
    return launch_fused_noise(audio, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, out, out_last, R, N, T, K, Lw, NJ,
                              delay_compensation, scale_kind, bias, exponent, max_value, threshold, gain,
                              voices_per_row, n_voices, voice_major, stream);
}

// ddspp_frequency_filter_eo_voices on noise that is DRAWN INSIDE THE KERNEL: exactly the U(-1, 1) numbers
// ddspp_uniform_noise(buf, R * N, seed, offset) would have written into a [R, N] tensor, which then never exists (config 3:
// 0.29 GB written and read back per step) -- DynamicSizeFilteredNoise.get_signal's tf.random.uniform feeding its
// frequency_filter (filtered_noise_synth.py:39-42) in one kernel.  Shapes: ddspp_frequency_filter_eo_drawn_supported.
int ddspp_frequency_filter_eo_drawn_supported(int N, int T, int K, int Lw, int delay_compensation) {
    WinGeom wg;
    return (!env_int("DDSPP_FIR_NO_FUSED", 0) && !env_int("DDSPP_NOISE_NO_DRAW", 0) &&
            win_fused_shape(N, T, K, Lw, delay_compensation, &wg)) ? 1 : 0;
}
int ddspp_frequency_filter_eo_voices_drawn(uint64_t seed, uint64_t offset, const float* magnitudes, const float* CE,
                                           const float* CO, const int* tap_idx, const float* tap_we, const float* tap_wo,
                                           float* out, float* out_last, int R, int N, int T, int K, int Lw, int NJ,
                                           int delay_compensation, int scale_kind, float bias, float exponent,
                                           float max_value, float threshold, float gain, int n_voices, int voices_per_row,
                                           int voice_major, hipStream_t stream) {
    return launch_fused_noise(nullptr, magnitudes, CE, CO, tap_idx, tap_we, tap_wo, out, out_last, R, N, T, K, Lw, NJ,
                              delay_compensation, scale_kind, bias, exponent, max_value, threshold, gain,
                              voices_per_row, n_voices, voice_major, stream, true, seed, offset);
}

// U(-1, 1) noise, Philox4x32-10(counter = offset + i / 4, key = seed); n % 4 == 0.
int ddspp_uniform_noise(float* out, size_t n, uint64_t seed, uint64_t offset, hipStream_t stream) {
    DDSPP_REQUIRE(out, "uniform_noise: null buffer");
    DDSPP_REQUIRE(n % 4 == 0, "uniform_noise: n must be a multiple of 4");
    if (n == 0) return DDSPP_OK;
    hipLaunchKernelGGL(uniform_noise_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, stream, out, n / 4, seed,
                       offset);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// The same generator for the rows of out[R, n] (n % 4 == 0), row r from the counters offset + r * row_stride + i / 4:
// a stream keyed by (row, position), so that a piece of a longer signal draws the numbers of its absolute position
// whichever call renders it (streaming.py; one launch for all rows).
int ddspp_uniform_noise_rows(float* out, int R, size_t n, uint64_t seed, uint64_t offset, uint64_t row_stride,
                             hipStream_t stream) {
    DDSPP_REQUIRE(out, "uniform_noise_rows: null buffer");
    DDSPP_REQUIRE(R >= 0 && n % 4 == 0, "uniform_noise_rows: n must be a multiple of 4");
    if (n == 0 || R == 0) return DDSPP_OK;
    hipLaunchKernelGGL(uniform_noise_rows_kernel, dim3(stream_grid((size_t)R * (n / 4))), dim3(256), 0, stream, out, R, n / 4,
                       seed, offset, row_stride);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

}  // extern "C"
