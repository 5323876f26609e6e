// Feedback-delay-network reverb: impulse-response generation (the step in front of
// the reverb on the maestro-v2 model).
//
// Replaces FeedbackDelayNetwork.get_late_ir (ddsp_piano/modules/fdn_reverb.py:178-334) as driven by
// get_ir (:336-360) and by MultiInstrumentFeedbackDelayReverb.call (sub_modules.py:431-446):
// the FDN is sampled at freq_points = 2 sr frequencies; per rfft bin an (D x D, D <= 8) complex
// system  (I - F diag(dd)) x = input_gain  is solved and projected,
//   H[n] = sum_i output_gain_i dd_i x_i,   F = diag(onepole) M diag(allpass),  dd = z^-floor(d) * interp,
// then late_ir = irfft(H) (rocFFT C2R, length 2 sr -- not a power of two).  One thread per
// (instrument, bin): the work is tiny (24001 bins x 8x8), control-side, and latency bound; complex64
// arithmetic in the reference's order, Gaussian elimination with partial pivoting instead of
// tf.linalg.inv + matmuls (same solution to round-off).
#include <rocfft/rocfft.h>

#include <mutex>

#include "ddspp_common.h"

namespace ddspp {

struct cf {
    float re, im;
};
__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cf cdiv(cf a, cf b) {
    const float d = b.re * b.re + b.im * b.im;
    return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
// Transcendentals (exp(j phi), 10^x) are evaluated in double and rounded once to float32: the correctly rounded
// float32 value, which every float32 libm (TF's Eigen, numpy, the GPU's) scatters around by an ulp or two.  Near the
// network's resonances cond(I - F D) ~ 1e4 turns one ulp of such scatter into 5e-4 of the response, so two
// implementations only agree when both use THE float32 value; the oracle (oracle/ddsp_oracle.py fdn_get_late_ir) does
// the same.  The arguments (wk * delay, -3 * delay / T60, ...) are formed in float32 exactly as the reference does.
__device__ __forceinline__ cf cexpi(float phi) {       // exp(j phi)
    double s, c;
    sincos((double)phi, &s, &c);
    return {(float)c, (float)s};
}
__device__ __forceinline__ float pow10_f32(float x) { return (float)pow(10.0, (double)x); }

struct cd {
    double re, im;
};
__device__ __forceinline__ cd dmul(cd a, cd b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cd dsub(cd a, cd b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cd ddiv(cd a, cd b) {
    const double d = b.re * b.re + b.im * b.im;
    return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}

constexpr int FDN_DMAX = 8;
constexpr int FDN_AMAX = 8;

struct FdnParams {
    const float* __restrict__ input_gain;      // [B, D]
    const float* __restrict__ output_gain;     // [B, D]
    const float* __restrict__ mixing;          // [D, D]
    const float* __restrict__ gain_allpass;    // [B, D, A]
    const float* __restrict__ delays_allpass;  // [B, D, A]
    const float* __restrict__ time_rev;        // [B]
    const float* __restrict__ alpha_tone;      // [B]
    const float* __restrict__ delay_values;    // [D]
    float2* __restrict__ H;                    // [B, nb]
    int B, D, A, freq_points, nb;
    float sr;
};

// SOLVE = 0: the D x D system of a bin is solved in float64 (default).  SOLVE = 1: the reference's own arithmetic,
// tf.linalg.inv in complex64 followed by the complex64 products (fdn_reverb.py:314-333) -- an LU inverse with partial
// pivoting, only good to cond(I - F D) x 6e-8 near the network's resonances; kept as a switch so that a TF-made golden
// can be matched either way.
template <int SOLVE>
__global__ void __launch_bounds__(256) fdn_transfer_kernel(const FdnParams p) {
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)p.B * p.nb) return;
    const int b = (int)(gid / p.nb), n = (int)(gid - (size_t)b * p.nb);
    const int D = p.D, A = p.A;
    // wk = 2*pi*n / freq_points in float32 (fdn_reverb.py:234-239)
    const float wk = (DDSPP_TWO_PI_F32 * (float)n) / (float)p.freq_points;
    const cf ez = cexpi(-wk);                                  // exp(-1j wk)
    const float t0 = p.time_rev[b], al = p.alpha_tone[b];

    cf dd[FDN_DMAX], filt[FDN_DMAX], ap[FDN_DMAX];
#pragma unroll
    for (int d = 0; d < FDN_DMAX; ++d) {
        if (d >= D) continue;
        const float dv = p.delay_values[d];
        const float fl = floorf(dv);
        const cf zd = cexpi(-(wk * fl));                       // z^-floor(d)               :241-250
        const float de = dv - fl;
        const float eta = (1.0f - de) / (1.0f + de);           //                            :253-254
        const cf interp = cdiv(cf{eta + ez.re, ez.im}, cf{1.0f + eta * ez.re, eta * ez.im});   // :255-261
        dd[d] = cmul(zd, interp);
        float dsum = 0.0f;
        for (int a = 0; a < A; ++a) dsum += p.delays_allpass[((size_t)b * D + d) * A + a];
        const float delay_sec = (dv + dsum) / p.sr;            //                            :265-268
        const float k = pow10_f32(-3.0f * delay_sec / t0);     //                            :272
        const float kpi = pow10_f32(-3.0f * delay_sec / (al * t0));
        const float g = 2.0f * k * kpi / (k + kpi);
        const float pp = (k - kpi) / (k + kpi);
        filt[d] = cdiv(cf{g, 0.0f}, cf{1.0f - pp * ez.re + 1e-8f, -pp * ez.im});   // g / (1 - p z + 1e-8)  :289-291
        cf prod = {1.0f, 0.0f};
        for (int a = 0; a < A; ++a) {
            const float ga = p.gain_allpass[((size_t)b * D + d) * A + a];
            const cf z = cexpi(wk * p.delays_allpass[((size_t)b * D + d) * A + a]);   // exp(+1j wk d)   :302
            prod = cmul(prod, cdiv(cf{1.0f + ga * z.re, ga * z.im}, cf{ga + z.re, z.im}));   // :305-308
        }
        ap[d] = prod;
    }
    if (SOLVE == 1) {
        // feedback = diag(filt) M diag(ap); a = I - feedback diag(dd); inv = a^-1 (Gauss-Jordan on [a | I], partial
        // pivoting); H = output_gain^T (diag(dd) inv) input_gain -- every product and sum in complex64
        cf a[FDN_DMAX][2 * FDN_DMAX];
#pragma unroll
        for (int i = 0; i < FDN_DMAX; ++i)
#pragma unroll
            for (int j = 0; j < FDN_DMAX; ++j) {
                cf v = {i == j ? 1.0f : 0.0f, 0.0f};
                if (i < D && j < D) {
                    const float m = p.mixing[i * D + j];
                    const cf f = cmul(cmul(cmul(filt[i], cf{m, 0.0f}), ap[j]), dd[j]);
                    v = csub(v, f);
                }
                a[i][j] = v;
                a[i][FDN_DMAX + j] = cf{i == j ? 1.0f : 0.0f, 0.0f};
            }
#pragma unroll
        for (int c = 0; c < FDN_DMAX; ++c) {
            float best = a[c][c].re * a[c][c].re + a[c][c].im * a[c][c].im;
            int piv = c;
#pragma unroll
            for (int r = c + 1; r < FDN_DMAX; ++r) {
                const float m = a[r][c].re * a[r][c].re + a[r][c].im * a[r][c].im;
                if (m > best) {
                    best = m;
                    piv = r;
                }
            }
#pragma unroll
            for (int r = c + 1; r < FDN_DMAX; ++r) {
                if (r == piv) {
#pragma unroll
                    for (int j = 0; j < 2 * FDN_DMAX; ++j) {
                        const cf t = a[c][j];
                        a[c][j] = a[r][j];
                        a[r][j] = t;
                    }
                }
            }
            const cf inv = cdiv(cf{1.0f, 0.0f}, a[c][c]);
#pragma unroll
            for (int j = 0; j < 2 * FDN_DMAX; ++j) a[c][j] = cmul(a[c][j], inv);
#pragma unroll
            for (int r = 0; r < FDN_DMAX; ++r) {
                if (r == c) continue;
                const cf f = a[r][c];
#pragma unroll
                for (int j = 0; j < 2 * FDN_DMAX; ++j) a[r][j] = csub(a[r][j], cmul(f, a[c][j]));
            }
        }
        cf h = {0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < FDN_DMAX; ++i) {
            if (i >= D) continue;
            cf row = {0.0f, 0.0f};
#pragma unroll
            for (int j = 0; j < FDN_DMAX; ++j)
                if (j < D) row = cadd(row, cmul(cmul(dd[i], a[i][FDN_DMAX + j]), cf{p.input_gain[(size_t)b * D + j], 0.0f}));
            h = cadd(h, cmul(cf{p.output_gain[(size_t)b * D + i], 0.0f}, row));
        }
        p.H[gid] = make_float2(h.re, h.im);
        return;
    }
    // a = I - F diag(dd),  F[i][j] = filt_i M_ij ap_j ; augmented with rhs = input_gain.
    // The solve runs in float64: I - F D is close to singular near the FDN's resonances (the reference's
    // own complex64 LU is only good to cond x 6e-8 there), and at 8 x 8 per bin double costs nothing.
    cd a[FDN_DMAX][FDN_DMAX + 1];
#pragma unroll
    for (int i = 0; i < FDN_DMAX; ++i) {
#pragma unroll
        for (int j = 0; j < FDN_DMAX; ++j) {
            cd v = {0.0, 0.0};
            if (i < D && j < D) {
                const double m = (double)p.mixing[i * D + j];
                const cd f = dmul(dmul(cd{filt[i].re * m, filt[i].im * m}, cd{ap[j].re, ap[j].im}), cd{dd[j].re, dd[j].im});
                v = cd{(i == j ? 1.0 : 0.0) - f.re, -f.im};
            } else if (i == j) {
                v = cd{1.0, 0.0};
            }
            a[i][j] = v;
        }
        a[i][FDN_DMAX] = (i < D) ? cd{(double)p.input_gain[(size_t)b * D + i], 0.0} : cd{0.0, 0.0};
    }
    // Gaussian elimination, partial pivoting (all indices static after unrolling; rows swapped by value)
#pragma unroll
    for (int c = 0; c < FDN_DMAX; ++c) {
        double best = a[c][c].re * a[c][c].re + a[c][c].im * a[c][c].im;
        int piv = c;
#pragma unroll
        for (int r = c + 1; r < FDN_DMAX; ++r) {
            const double m = a[r][c].re * a[r][c].re + a[r][c].im * a[r][c].im;
            if (m > best) {
                best = m;
                piv = r;
            }
        }
#pragma unroll
        for (int r = c + 1; r < FDN_DMAX; ++r) {
            if (r == piv) {
#pragma unroll
                for (int j = 0; j <= FDN_DMAX; ++j) {
                    const cd t = a[c][j];
                    a[c][j] = a[r][j];
                    a[r][j] = t;
                }
            }
        }
        const cd inv = ddiv(cd{1.0, 0.0}, a[c][c]);
#pragma unroll
        for (int r = c + 1; r < FDN_DMAX; ++r) {
            const cd f = dmul(a[r][c], inv);
#pragma unroll
            for (int j = c + 1; j <= FDN_DMAX; ++j) a[r][j] = dsub(a[r][j], dmul(f, a[c][j]));
        }
    }
    cd x[FDN_DMAX];
#pragma unroll
    for (int i = FDN_DMAX - 1; i >= 0; --i) {
        cd s = a[i][FDN_DMAX];
#pragma unroll
        for (int j = i + 1; j < FDN_DMAX; ++j) s = dsub(s, dmul(a[i][j], x[j]));
        x[i] = ddiv(s, a[i][i]);
    }
    cd h = {0.0, 0.0};
#pragma unroll
    for (int i = 0; i < FDN_DMAX; ++i) {
        if (i < D) {
            const double og = (double)p.output_gain[(size_t)b * D + i];
            const cd t = dmul(cd{dd[i].re, dd[i].im}, x[i]);
            h = cd{h.re + og * t.re, h.im + og * t.im};
        }
    }
    p.H[gid] = make_float2((float)h.re, (float)h.im);
}

// ir[b, i] = late[b, i] + (i < E ? early[b, i] : 0)           fdn_reverb.py:354-360
__global__ void __launch_bounds__(256) fdn_add_early_kernel(float* __restrict__ ir, const float* __restrict__ early,
                                                          int B, int L, int E) {
    const size_t total = (size_t)B * min(E, L);
    const int e = min(E, L);
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int b = (int)(g / e), i = (int)(g - (size_t)b * e);
        ir[(size_t)b * L + i] += early[(size_t)b * E + i];
    }
}

struct C2rPlan {
    int n, batch;
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    size_t work_bytes = 0;
};

static std::once_flag g_fdn_rocfft_once;

}  // namespace ddspp

using namespace ddspp;

#define DDSPP_FFT_CHECK(expr)                                                              \
    do {                                                                                   \
        rocfft_status _s = (expr);                                                         \
        if (_s != rocfft_status_success) {                                                 \
            ddspp_set_error("%s failed with rocfft_status %d (%s:%d)", #expr, (int)_s,     \
                            __FILE__, __LINE__);                                           \
            return DDSPP_EFFT;                                                             \
        }                                                                                  \
    } while (0)

extern "C" {

typedef struct C2rPlan ddspp_irfft_plan;

// tf.signal.irfft of length n (any n), `batch` rows of n / 2 + 1 complex64 bins -> n float32, 1/n scaled
int ddspp_irfft_plan_create(int n, int batch, ddspp_irfft_plan** out_plan) {
    DDSPP_REQUIRE(out_plan && n >= 2 && n % 2 == 0 && batch > 0, "irfft_plan_create: bad arguments");
    std::call_once(g_fdn_rocfft_once, [] { rocfft_setup(); });
    C2rPlan* pl = new C2rPlan();
    pl->n = n;
    pl->batch = batch;
    rocfft_plan_description desc = nullptr;
    DDSPP_FFT_CHECK(rocfft_plan_description_create(&desc));
    DDSPP_FFT_CHECK(rocfft_plan_description_set_scale_factor(desc, 1.0 / (double)n));
    const size_t lengths[1] = {(size_t)n};
    DDSPP_FFT_CHECK(rocfft_plan_create(&pl->plan, rocfft_placement_notinplace, rocfft_transform_type_real_inverse,
                                       rocfft_precision_single, 1, lengths, (size_t)batch, desc));
    rocfft_plan_description_destroy(desc);
    DDSPP_FFT_CHECK(rocfft_plan_get_work_buffer_size(pl->plan, &pl->work_bytes));
    DDSPP_FFT_CHECK(rocfft_execution_info_create(&pl->info));
    *out_plan = pl;
    return DDSPP_OK;
}

int ddspp_irfft_plan_destroy(ddspp_irfft_plan* pl) {
    if (!pl) return DDSPP_OK;
    if (pl->plan) rocfft_plan_destroy(pl->plan);
    if (pl->info) rocfft_execution_info_destroy(pl->info);
    delete pl;
    return DDSPP_OK;
}

size_t ddspp_irfft_workspace_bytes(const ddspp_irfft_plan* pl) { return pl ? pl->work_bytes : 0; }

// spectrum (device, [batch, n/2+1] complex64, DESTROYED) -> signal (device, [batch, n] float32)
int ddspp_irfft_execute(ddspp_irfft_plan* pl, void* spectrum, float* signal, void* workspace,
                        size_t workspace_bytes, hipStream_t stream) {
    DDSPP_REQUIRE(pl && spectrum && signal, "irfft_execute: null argument");
    DDSPP_REQUIRE(workspace_bytes >= pl->work_bytes && (pl->work_bytes == 0 || workspace),
                  "irfft_execute: workspace too small");
    DDSPP_FFT_CHECK(rocfft_execution_info_set_stream(pl->info, stream));
    if (pl->work_bytes) DDSPP_FFT_CHECK(rocfft_execution_info_set_work_buffer(pl->info, workspace, pl->work_bytes));
    void* in[1] = {spectrum};
    void* out[1] = {signal};
    DDSPP_FFT_CHECK(rocfft_execute(pl->plan, in, out, pl->info));
    return DDSPP_OK;
}

// FeedbackDelayNetwork.get_late_ir up to (not including) the irfft -- fdn_reverb.py:178-334, for B
// instruments at once (the tf.vectorized_map of sub_modules.py:444).  H: [B, freq_points/2 + 1] complex64.
// solve: DDSPP_FDN_SOLVE_F64 (0) or DDSPP_FDN_SOLVE_C64_INVERSE (1, the reference's tf.linalg.inv arithmetic).
int ddspp_fdn_transfer(const float* input_gain, const float* output_gain, const float* mixing_matrix,
                       const float* gain_allpass, const float* delays_allpass, const float* time_rev_0_sec,
                       const float* alpha_tone, const float* delay_values, void* H, int B, int D, int A,
                       int freq_points, float sampling_rate, int solve, hipStream_t stream) {
    DDSPP_REQUIRE(input_gain && output_gain && mixing_matrix && gain_allpass && delays_allpass && time_rev_0_sec &&
                      alpha_tone && delay_values && H, "fdn_transfer: null buffer");
    DDSPP_REQUIRE(B > 0 && D > 0 && D <= FDN_DMAX && A > 0 && A <= FDN_AMAX && freq_points >= 2 && freq_points % 2 == 0,
                  "fdn_transfer: bad dims (delay_lines <= 8, allpass stages <= 8)");
    DDSPP_REQUIRE(solve == DDSPP_FDN_SOLVE_F64 || solve == DDSPP_FDN_SOLVE_C64_INVERSE, "fdn_transfer: unknown solve mode %d", solve);
    FdnParams p{};
    p.input_gain = input_gain; p.output_gain = output_gain; p.mixing = mixing_matrix;
    p.gain_allpass = gain_allpass; p.delays_allpass = delays_allpass; p.time_rev = time_rev_0_sec;
    p.alpha_tone = alpha_tone; p.delay_values = delay_values; p.H = (float2*)H;
    p.B = B; p.D = D; p.A = A; p.freq_points = freq_points; p.nb = freq_points / 2 + 1; p.sr = sampling_rate;
    const size_t total = (size_t)B * p.nb;
    if (solve == DDSPP_FDN_SOLVE_C64_INVERSE)
        hipLaunchKernelGGL(fdn_transfer_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(fdn_transfer_kernel<0>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

// ir = late_ir + zero-padded early_ir  (FeedbackDelayNetwork.get_ir, fdn_reverb.py:354-360)
int ddspp_fdn_add_early(float* ir, const float* early_ir, int B, int L, int E, hipStream_t stream) {
    DDSPP_REQUIRE(ir && early_ir && B > 0 && L > 0 && E > 0, "fdn_add_early: bad arguments");
    const size_t total = (size_t)B * (E < L ? E : L);
    hipLaunchKernelGGL(fdn_add_early_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ir,
                       early_ir, B, L, E);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

}  // extern "C"
