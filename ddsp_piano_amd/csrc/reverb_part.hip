// Long convolution reverb as a uniformly partitioned overlap-save convolution with the FFT blocks resident in LDS.
//
// Same operator as reverb.hip -- ddsp.core.fft_convolve, single impulse-response frame, as called by
// ddsp.effects.Reverb.get_signal and FeedbackDelayNetwork.get_signal (ddsp_piano/modules/fdn_reverb.py:407-410):
//   out[n] = (audio * ir)[n + start]  (+ audio[n])
// The reference (and reverb.hip's rocFFT route) transforms the whole zero-padded signal at once: at 3 s + 3 s that is a
// 163 840-point transform per row whose passes, paddings, product and crop cross HBM thirteen times (0.97 GB moved for
// 55 MB of audio, impulse response and output).  A linear convolution does not care how it is cut:
//   y[i Bs + t] = sum_p (x window i - p) (*) (h partition p),     Bs = 4096,
// with 8192-point real transforms that fit the LDS of a CU.  Four launches:
//   part_fwd_kernel<IR>   : h partitions -> spectra H[row, p, :]           (on the caller's side stream, early)
//   part_fwd_kernel<AUDIO>: x windows    -> spectra X[row, j, :]           (reads the dry mix once, from HBM / L2)
//   part_mac_kernel       : Y[row, i, :] = sum_p X[i - p] H[p], bin-major: a thread owns a bin, eight output blocks'
//                           accumulators and a sliding window of X in registers (one X and one H load per 8 products)
//   part_inv_kernel       : Y -> inverse transform in LDS, crop, + dry -> out
// HBM sees the audio, the impulse responses, the output and three sets of spectra written once and read back a few
// times (mostly from L2): a fraction of the traffic of the whole-signal route.
//
// The 8192-point real transform is a 4096-point complex one on (even, odd) sample pairs plus an unpack step; the
// 4096-point transform is three radix-16 Stockham passes by 256 threads (one 16-point butterfly per thread and pass),
// twiddles from a table built by the host in double precision.  The Nyquist bin rides in imag(bin 0).
#include "ddspp_common.h"
#include "reverb_part.h"

namespace ddspp {

namespace {

constexpr int PM = REVERB_PART_BLOCK;          // 4096: complex transform size = hop in real samples
constexpr int PTH = 256;                       // threads; PM / 16
constexpr int PLDS = PM + PM / 16;             // padded: element idx lives at idx + idx / 16 (conflict-free passes)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ int pidx(int i) { return i + (i >> 4); }

// 4-point DFT in place, forward (W_4 = -i) or inverse (+i)
template <bool INV>
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 b0 = cadd(a0, a2), b1 = csub(a0, a2), b2 = cadd(a1, a3), b3 = csub(a1, a3);
    const float2 ib3 = INV ? make_float2(-b3.y, b3.x) : make_float2(b3.y, -b3.x);      // (+/-) i b3 ... forward: -i b3
    a0 = cadd(b0, b2);
    a2 = csub(b0, b2);
    a1 = cadd(b1, ib3);
    a3 = csub(b1, ib3);
}

// 16-point DFT of v (natural order in, natural order out): n = 4 n1 + n2, k = k1 + 4 k2
template <bool INV>
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
    float2 t[4][4];                            // t[n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        float2 a0 = v[n2], a1 = v[4 + n2], a2 = v[8 + n2], a3 = v[12 + n2];
        dft4<INV>(a0, a1, a2, a3);
        t[n2][0] = a0; t[n2][1] = a1; t[n2][2] = a2; t[n2][3] = a3;
    }
    // W_16^(n2 k1): m = 1, 2, 3, 2, 4, 6, 3, 6, 9 (forward exp(-2 pi i m / 16), inverse its conjugate)
    const float sg = INV ? 1.0f : -1.0f;
    const float2 w1 = make_float2(C1, sg * S1), w2 = make_float2(R2, sg * R2), w3 = make_float2(S1, sg * C1);
    const float2 w4 = make_float2(0.0f, sg), w6 = make_float2(-R2, sg * R2), w9 = make_float2(-C1, -sg * S1);
    t[1][1] = cmul(t[1][1], w1); t[1][2] = cmul(t[1][2], w2); t[1][3] = cmul(t[1][3], w3);
    t[2][1] = cmul(t[2][1], w2); t[2][2] = cmul(t[2][2], w4); t[2][3] = cmul(t[2][3], w6);
    t[3][1] = cmul(t[3][1], w3); t[3][2] = cmul(t[3][2], w6); t[3][3] = cmul(t[3][3], w9);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        float2 a0 = t[0][k1], a1 = t[1][k1], a2 = t[2][k1], a3 = t[3][k1];
        dft4<INV>(a0, a1, a2, a3);
        v[k1] = a0; v[k1 + 4] = a1; v[k1 + 8] = a2; v[k1 + 12] = a3;
    }
}

// Twiddles of passes 2 and 3 for thread j: fetched at kernel start (they do not depend on the data), so that the passes
// do not wait for the table.  W = exp(-2 pi i e / 4096).
struct Twiddles {
    float2 w2[15], w3[15];
};
__device__ __forceinline__ void load_twiddles(Twiddles& t, const float2* __restrict__ W, int j) {
    const int k = j & 15;
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        t.w2[r - 1] = W[16 * k * r];           // W_256^(k r)
        t.w3[r - 1] = W[j * r];                // W_4096^(j r)
    }
}

// The three Stockham passes.  On entry v[r] = input[j + 256 r] (registers); on exit the transform sits in `buf`
// (padded, natural order) and the workgroup is synchronised.
template <bool INV>
__device__ __forceinline__ void fft4096(float2 (&v)[16], float2* __restrict__ buf, const Twiddles& tw, int j) {
    // pass 1: Ns = 1, no twiddles; out[16 j + r]
    dft16<INV>(v);
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[pidx(16 * j + r)] = v[r];
    __syncthreads();
    // pass 2: Ns = 16
    {
        const int k = j & 15;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = buf[pidx(j + 256 * r)];
#pragma unroll
        for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], INV ? cconj(tw.w2[r - 1]) : tw.w2[r - 1]);
        dft16<INV>(v);
        __syncthreads();
        const int base = (j >> 4) * 256 + k;
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[pidx(base + 16 * r)] = v[r];
        __syncthreads();
    }
    // pass 3: Ns = 256; out[j + 256 r]
    {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = buf[pidx(j + 256 * r)];
#pragma unroll
        for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], INV ? cconj(tw.w3[r - 1]) : tw.w3[r - 1]);
        dft16<INV>(v);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[pidx(j + 256 * r)] = v[r];
        __syncthreads();
    }
}

// Spectrum of one 8192-sample window per workgroup.  IR: window = [h[p Bs .. (p + 1) Bs), zeros] (h[0] masked when
// mask_first); AUDIO: window = x[(jb - 1) Bs .. (jb + 1) Bs), zero outside the signal.
template <bool IR>
__global__ void __launch_bounds__(PTH) part_fwd_kernel(const float* __restrict__ src, int src_stride, int n_src,
                                                      float2* __restrict__ spec, int nblk, int mask_first,
                                                      const float2* __restrict__ W, const float2* __restrict__ U) {
    __shared__ float2 buf[PLDS];
    const int j = threadIdx.x;
    const int row = blockIdx.x / nblk, b = blockIdx.x - row * nblk;
    const float* s = src + (size_t)row * src_stride;
    const long long lo = IR ? (long long)b * PM : ((long long)b - 1) * PM;
    Twiddles tw;
    load_twiddles(tw, W, j);
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = j + 256 * r;                 // complex index: real samples 2 n, 2 n + 1 of the window
        const long long t0 = lo + 2 * n;
        float a = 0.0f, c = 0.0f;
        if (!IR || 2 * n < PM) {
            if (t0 >= 0 && t0 < n_src) a = s[t0];
            if (t0 + 1 >= 0 && t0 + 1 < n_src) c = s[t0 + 1];
            if (IR && mask_first && t0 == 0) a = 0.0f;               // ddsp.effects.Reverb._mask_dry_ir
        }
        v[r] = make_float2(a, c);
    }
    fft4096<false>(v, buf, tw, j);
    // unpack: bins k and M - k from Z[k], Z[M - k]
    float2* out = spec + ((size_t)row * nblk + b) * PM;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = j + 256 * q;                 // 0 .. 2047
        const float2 zk = buf[pidx(k)];
        if (k == 0) {
            out[0] = make_float2(zk.x + zk.y, zk.x - zk.y);          // DC, Nyquist (both real) share bin 0
            const float2 zh = buf[pidx(PM / 2)];
            out[PM / 2] = cconj(zh);
        } else {
            const float2 zm = buf[pidx(PM - k)];
            const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));           // (Zk + conj Zm) / 2
            const float2 d = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));           // (Zk - conj Zm) / 2
            const float2 o = make_float2(d.y, -d.x);                                           // -i d
            const float2 uo = cmul(U[k], o);
            out[k] = cadd(e, uo);
            out[PM - k] = cconj(csub(e, uo));
        }
    }
}

// Y[row, i, :] = sum_p X[row, i - p, :] H[row, p, :] for a chunk of MAC_CH consecutive output blocks and 256 bins per
// workgroup.  A thread owns one bin: the chunk's accumulators and a sliding window of the X spectra stay in registers,
// a step of the p loop loads ONE new X value and ONE H value for MAC_CH products -- every spectrum is read (8 + Pn) / 8
// times per chunk instead of once per output block (the first version of this file read 0.65 GB here).
constexpr int MAC_CH = 8;
constexpr int MAC_AHEAD = 4;      // steps of the p loop whose X and H values are already requested
// (round 4: the loop used to load H[p] and use it in the next instruction, and to take its new X value behind the products
// -- one exposed memory latency per step of a loop whose arithmetic is eight complex multiply-adds; and every product
// carried both the complex and the "bin 0 holds two real bins" form under EXEC masks: 248 EXEC branches in the kernel.
// Now the values of MAC_AHEAD steps ahead are in flight, a product is four FMAs, and the packed bin is patched up by the
// one wavefront that owns it.)
template <bool PACKED>
__device__ __forceinline__ void mac_step(float2 (&acc)[MAC_CH], const float2 (&win)[MAC_CH], int u, float2 h, bool packed) {
#pragma unroll
    for (int c = 0; c < MAC_CH; ++c) {
        const float2 x = win[(c - u + MAC_CH) % MAC_CH];       // window slot (c - u) mod MAC_CH holds X_{ic + c - p}
        float re = __builtin_fmaf(x.x, h.x, acc[c].x), im = __builtin_fmaf(x.x, h.y, acc[c].y);
        const float re2 = __builtin_fmaf(-x.y, h.y, re), im2 = __builtin_fmaf(x.y, h.x, im);
        if (PACKED && packed) {                                  // bin 0 holds (DC, Nyquist): two real products
            im = __builtin_fmaf(x.y, h.y, acc[c].y);
            acc[c] = make_float2(re, im);
        } else {
            acc[c] = make_float2(re2, im2);
        }
    }
}

__global__ void __launch_bounds__(256) part_mac_kernel(const float2* __restrict__ X, const float2* __restrict__ H,
                                                     float2* __restrict__ Y, int nbx, int Pn, int hrow_stride_blocks, int i0,
                                                     int nbo, int jmax, int nchunks) {
    const int tiles = PM / 256;
    int id = blockIdx.x;
    const int tile = id % tiles; id /= tiles;
    const int chunk = id % nchunks;
    const int row = id / nchunks;
    const int bin = tile * 256 + threadIdx.x;
    const int ic = i0 + chunk * MAC_CH;                       // first output block of the chunk
    const float2* xrow = X + (size_t)row * nbx * PM + bin;
    const float2* hrow = H + (size_t)row * hrow_stride_blocks * PM + bin;
    auto xload = [&](int jb) {                                // X_j is zero outside [0, jmax] (jb is wave-uniform)
        const float2 v = xrow[(size_t)min(max(jb, 0), jmax) * PM];
        return (jb >= 0 && jb <= jmax) ? v : make_float2(0.0f, 0.0f);
    };
    float2 acc[MAC_CH], win[MAC_CH];
#pragma unroll
    for (int c = 0; c < MAC_CH; ++c) {
        acc[c] = make_float2(0.0f, 0.0f);
        win[c] = xload(ic + c);                               // p = 0: X_{ic + c}
    }
    const bool packed = bin == 0;
    const bool packed_wave = wave_uniform(tile == 0 && threadIdx.x < 64);
    const int p_end = min(Pn - 1, ic + MAC_CH - 1);           // beyond, i - p < 0 for every i of the chunk
    // hq[k] / xq[k]: H_p and the X value that enters the window after step p, for the steps p = p_cur + k
    float2 hq[MAC_AHEAD], xq[MAC_AHEAD];
#pragma unroll
    for (int k = 0; k < MAC_AHEAD; ++k) {
        hq[k] = hrow[(size_t)min(k, p_end) * PM];
        xq[k] = xload(ic - k - 1);
    }
    for (int p0 = 0; p0 <= p_end; p0 += MAC_CH) {
#pragma unroll
        for (int u = 0; u < MAC_CH; ++u) {
            const int p = p0 + u;
            if (p > p_end) break;
            const float2 h = hq[u % MAC_AHEAD], xin = xq[u % MAC_AHEAD];
            hq[u % MAC_AHEAD] = hrow[(size_t)min(p + MAC_AHEAD, p_end) * PM];
            xq[u % MAC_AHEAD] = xload(ic - (p + MAC_AHEAD) - 1);
            if (packed_wave) mac_step<true>(acc, win, u, h, packed);
            else mac_step<false>(acc, win, u, h, false);
            // next p: every slot's block index drops by one; the slot that held X_{ic + MAC_CH - 1 - p} takes X_{ic - p - 1}
            win[(MAC_CH - 1 - u + MAC_CH) % MAC_CH] = xin;
        }
    }
#pragma unroll
    for (int c = 0; c < MAC_CH; ++c) {
        const int i = ic + c;
        if (i < i0 + nbo) Y[((size_t)row * nbo + (i - i0)) * PM + bin] = acc[c];
    }
}

// Output block i of one row per workgroup: Y -> inverse transform in LDS, crop, + dry.
__global__ void __launch_bounds__(PTH) part_inv_kernel(const float2* __restrict__ Y, const float* __restrict__ dry,
                                                     int dry_stride, float* __restrict__ out, int out_len, int start, int i0,
                                                     int nbo, const float2* __restrict__ W, const float2* __restrict__ U) {
    __shared__ float2 buf[PLDS];
    const int j = threadIdx.x;
    const int row = blockIdx.x / nbo, ib = blockIdx.x - row * nbo, i = i0 + ib;
    Twiddles tw;
    load_twiddles(tw, W, j);
    const float2* yrow = Y + ((size_t)row * nbo + ib) * PM;
    float2 acc[16], uu[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = yrow[j + 256 * r];
    // (round 4: the unpack factors and the dry samples are requested HERE, with the spectrum -- they used to be loaded
    // where they are used, sixteen + sixteen loads each followed by its own wait.  A workgroup that walks several blocks
    // with the next block's spectrum in flight was tried as well: 62 us against 39, the registers do not fit.)
#pragma unroll
    for (int r = 0; r < 16; ++r) uu[r] = U[j + 256 * r];
    const float* drow = dry ? dry + (size_t)row * dry_stride : nullptr;
    float dv[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const long long m0 = (long long)i * PM + 2 * (j + 256 * q) - start;
        dv[2 * q] = dv[2 * q + 1] = 0.0f;
        if (drow) {                                                                    // (uniform)
            const float a = drow[min(max(m0, 0ll), (long long)out_len - 1)], c = drow[min(max(m0 + 1, 0ll), (long long)out_len - 1)];
            dv[2 * q] = (m0 >= 0 && m0 < out_len) ? a : 0.0f;
            dv[2 * q + 1] = (m0 + 1 >= 0 && m0 + 1 < out_len) ? c : 0.0f;
        }
    }
    // Y -> Z' = E + i O,  E = (Y[k] + conj Y[M - k]) / 2,  O = (Y[k] - conj Y[M - k]) / 2 * conj U[k]
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[pidx(j + 256 * r)] = acc[r];
    __syncthreads();
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = j + 256 * r;
        float2 yk = acc[r], ym;
        if (k == 0) {
            ym = make_float2(yk.y, 0.0f);                    // conj(Y[M]) = Nyquist
            yk = make_float2(yk.x, 0.0f);                    // DC
        } else {
            ym = cconj(buf[pidx(PM - k)]);
        }
        const float2 e = make_float2(0.5f * (yk.x + ym.x), 0.5f * (yk.y + ym.y));
        const float2 d = make_float2(0.5f * (yk.x - ym.x), 0.5f * (yk.y - ym.y));
        const float2 o = cmul(d, cconj(uu[r]));
        v[r] = make_float2(e.x - o.y, e.y + o.x);            // e + i o
    }
    __syncthreads();
    fft4096<true>(v, buf, tw, j);
    // the last Bs of the 2 Bs outputs: complex n in [M / 2, M) -> real samples 2 (n - M / 2), + 1 of block i
    const float scale = 1.0f / (float)PM;
    float* orow = out + (size_t)row * out_len;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int n = PM / 2 + j + 256 * q;
        const float2 z = buf[pidx(n)];
        const long long m0 = (long long)i * PM + 2 * (n - PM / 2) - start;      // output index of the real part
        if (m0 >= 0 && m0 < out_len) orow[m0] = z.x * scale + dv[2 * q];
        if (m0 + 1 >= 0 && m0 + 1 < out_len) orow[m0 + 1] = z.y * scale + dv[2 * q + 1];
    }
}

}  // namespace

void reverb_part_tables_host(float* W, float* U) {
    const double two_pi = 6.283185307179586476925286766559;
    for (int e = 0; e < PM; ++e) {
        W[2 * e] = (float)cos(two_pi * e / PM);
        W[2 * e + 1] = (float)-sin(two_pi * e / PM);
        U[2 * e] = (float)cos(two_pi * e / (2.0 * PM));
        U[2 * e + 1] = (float)-sin(two_pi * e / (2.0 * PM));
    }
}

int reverb_part_transform_ir(const PartPlan& pp, const float* ir, int B_ir, int L, int mask_dry, float2* Hspec,
                             hipStream_t stream) {
    hipLaunchKernelGGL((part_fwd_kernel<true>), dim3((unsigned)(B_ir * pp.Pn)), dim3(PTH), 0, stream, ir, L, L, Hspec, pp.Pn,
                       mask_dry, pp.W, pp.U);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

int reverb_part_execute(const PartPlan& pp, const float* audio, int audio_stride, int B, int B_ir, int N, float2* Xspec,
                        const float2* Hspec, float2* Yspec, float* out, int out_len, int start, int add_dry, hipStream_t stream) {
    const int nbo_all = (start + out_len + PM - 1) / PM;           // output blocks 0 .. nbo_all - 1 cover y[0, start + out_len)
    const int i0 = start / PM;
    int jmax = (N + PM - 1) / PM;                                  // X_j is zero for j > ceil(N / Bs)
    if (jmax > nbo_all - 1) jmax = nbo_all - 1;
    DDSPP_REQUIRE(jmax + 1 <= pp.nbx, "fft_convolve (partitioned): %d audio blocks exceed the plan's %d", jmax + 1, pp.nbx);
    hipLaunchKernelGGL((part_fwd_kernel<false>), dim3((unsigned)(B * (jmax + 1))), dim3(PTH), 0, stream, audio, audio_stride,
                       N, Xspec, jmax + 1, 0, pp.W, pp.U);
    DDSPP_LAUNCH_CHECK();
    const int nbo = nbo_all - i0;
    DDSPP_REQUIRE(nbo <= pp.nbo, "fft_convolve (partitioned): %d output blocks exceed the plan's %d", nbo, pp.nbo);
    const int nchunks = (nbo + MAC_CH - 1) / MAC_CH;
    hipLaunchKernelGGL(part_mac_kernel, dim3((unsigned)(B * nchunks * (PM / 256))), dim3(256), 0, stream, Xspec, Hspec, Yspec,
                       jmax + 1, pp.Pn, B_ir == 1 ? 0 : pp.Pn, i0, nbo, jmax, nchunks);
    DDSPP_LAUNCH_CHECK();
    hipLaunchKernelGGL(part_inv_kernel, dim3((unsigned)(B * nbo)), dim3(PTH), 0, stream, Yspec, add_dry ? audio : nullptr,
                       audio_stride, out, out_len, start, i0, nbo, pp.W, pp.U);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

}  // namespace ddspp
