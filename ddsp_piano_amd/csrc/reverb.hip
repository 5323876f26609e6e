// Long convolution reverb through rocFFT.
//
// Replaces ddsp.core.fft_convolve for the single-frame case (one impulse response per batch row)
// as called by ddsp.effects.Reverb.get_signal (dry-masked IR, delay_compensation=0, + dry) and by
// FeedbackDelayNetwork.get_signal (ddsp_piano/modules/fdn_reverb.py:407-410).
//   fft_size = 2 ** ceil(log2(N + L - 1))            ddsp.core.get_fft_size(power_of_2=True); here the cheapest
//                                                     of 2^k, 3 * 2^k, 5 * 2^k, ... >= N + L - 1 (fast_fft_size)
//   out[n]   = irfft(rfft(audio, fft_size) * rfft(ir, fft_size))[n + delay]  (+ audio[n])
// rocFFT owns the butterflies; the hand-written kernels around it fuse the dry-sample mask into the
// zero padding of the IR, and the crop + add-dry into one epilogue pass.  The library owns only the
// rocFFT plans behind the opaque handle; every buffer (including rocFFT's work buffer) belongs to
// the caller.
#include <rocfft/rocfft.h>

#include <mutex>

#include "ddspp_common.h"
#include "reverb_part.h"

namespace ddspp {

struct FftConvPlan {
    int B, B_ir, N, L, nfft;
    // partitioned overlap-save route (reverb_part.hip): used when the signal spans several 4096-sample blocks
    bool part = false;
    PartPlan pp{};
    float2* tables = nullptr;            // device: W then U (the only device memory the plan owns: 64 KB)
    size_t off_xspec = 0, off_hspec = 0, off_yspec = 0;
    rocfft_plan fwd_audio = nullptr, fwd_ir = nullptr, inv = nullptr;
    rocfft_execution_info info = nullptr;
    size_t rocfft_work_bytes = 0;
    // workspace layout (bytes, 256-aligned)
    size_t off_audio_p, off_ir_p, off_audio_f, off_ir_f, off_work, total_bytes;
};

static std::once_flag g_rocfft_once;
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// dst[b, i] = i < n_src ? src[b, i] : 0 ; optionally dst[b, 0] = 0   (Reverb._mask_dry_ir)
__global__ void __launch_bounds__(256) pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                     int rows, int n_src, int src_stride, int n_dst,
                                                     int mask_first) {
    const int q = n_dst / 4;
    const size_t total = (size_t)rows * q;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int b = (int)(g / q), i = (int)(g - (size_t)b * q) * 4;
        const float* s = src + (size_t)b * src_stride;
        float4 v;
        v.x = (i + 0 < n_src) ? s[i + 0] : 0.0f;
        v.y = (i + 1 < n_src) ? s[i + 1] : 0.0f;
        v.z = (i + 2 < n_src) ? s[i + 2] : 0.0f;
        v.w = (i + 3 < n_src) ? s[i + 3] : 0.0f;
        if (mask_first && i == 0) v.x = 0.0f;
        reinterpret_cast<float4*>(dst + (size_t)b * n_dst)[i / 4] = v;
    }
}

// A[b, k] *= H[b % B_ir.., k]   (complex, interleaved)
__global__ void __launch_bounds__(256) spectrum_multiply_kernel(float2* __restrict__ a,
                                                              const float2* __restrict__ h, int B,
                                                              int B_ir, int nbins) {
    const size_t total = (size_t)B * nbins;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int b = (int)(g / nbins), k = (int)(g - (size_t)b * nbins);
        const float2 x = a[g];
        const float2 y = h[(size_t)(B_ir == 1 ? 0 : b) * nbins + k];
        float2 o;
        o.x = x.x * y.x - x.y * y.y;
        o.y = x.x * y.y + x.y * y.x;
        a[g] = o;
    }
}

// out[b, n] = y[b, n + delay] (+ dry[b, n])      crop_and_compensate_delay + Reverb add_dry
__global__ void __launch_bounds__(256) crop_add_dry_kernel(const float* __restrict__ y,
                                                         const float* __restrict__ dry,
                                                         float* __restrict__ out, int B, int out_len,
                                                         int nfft, int delay, int dry_stride) {
    const size_t total = (size_t)B * out_len;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const int b = (int)(g / out_len), n = (int)(g - (size_t)b * out_len);
        float v = y[(size_t)b * nfft + n + delay];
        if (dry) v = v + dry[(size_t)b * dry_stride + n];
        out[g] = v;
    }
}

static unsigned stream_grid(size_t total) {
    size_t blocks = (total + 255) / 256;
    const size_t cap = 256 * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

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

typedef struct FftConvPlan ddspp_fftconv_plan;

// ddsp.core.get_fft_size(frame_size=N, ir_size=L, power_of_2=True)
int ddspp_fft_size(int N, int L) {
    long long need = (long long)N + L - 1;
    long long n = 1;
    while (n < need) n <<= 1;
    return n > 0x40000000ll ? -1 : (int)n;
}

// Transform length the plan really uses.  The reference pads to the next power of two (ddspp_fft_size); the
// linear convolution does not depend on the padding, so any length >= N + L - 1 that rocFFT handles well will do.
// Measured on MI355X (64 rows, N = L = 72000, whole fft_convolve): 262144 = 2^18 0.389 ms, 163840 = 5 * 2^15
// 0.280 ms, 196608 = 3 * 2^16 0.348 ms, 147456 = 9 * 2^14 0.346 ms, 144000 = 2^7 3^2 5^3 0.346 ms: per point the
// power of two is cheapest, {3, 5} * 2^k cost ~1.18x, other 5-smooth lengths ~1.6x.  The smallest such cost wins.
// DDSPP_FFT_POW2=1 keeps the reference's size, DDSPP_FFT_SIZE=n forces a length (tuning).
static int fast_fft_size(int N, int L) {
    const long long need = (long long)N + L - 1;
    if (ddspp_option_literal("DDSPP_FFT_POW2", 0) == 1) return ddspp_fft_size(N, L);
    const int forced = ddspp_option_literal("DDSPP_FFT_SIZE", 0);
    if (forced >= need && forced % 8 == 0) return forced;
    long long best = -1;
    double best_cost = 0.0;
    for (int p5 = 0; p5 <= 3; ++p5)
        for (int p3 = 0; p3 <= 2; ++p3) {
            long long v = 1;
            for (int i = 0; i < p5; ++i) v *= 5;
            for (int i = 0; i < p3; ++i) v *= 3;
            while (v < need || v % 8 != 0) v *= 2;
            const double weight = (p3 + p5 == 0) ? 1.0 : (p3 + p5 == 1 ? 1.18 : 1.6);
            const double cost = weight * (double)v;
            if (best < 0 || cost < best_cost) {
                best = v;
                best_cost = cost;
            }
        }
    return (best < 8 || best > 0x40000000ll) ? -1 : (int)best;
}

int ddspp_fftconv_plan_create(int B, int B_ir, int N, int L, ddspp_fftconv_plan** out_plan) {
    DDSPP_REQUIRE(out_plan, "fftconv_plan_create: null out_plan");
    DDSPP_REQUIRE(B > 0 && N > 0 && L > 0, "fftconv_plan_create: bad dims");
    DDSPP_REQUIRE(B_ir == B || B_ir == 1,
                  "Batch size of audio (%d) and impulse response (%d) must be the same.", B, B_ir);
    const int nfft = fast_fft_size(N, L);
    DDSPP_REQUIRE(nfft >= 8, "fftconv_plan_create: fft size out of range");
    FftConvPlan* pl = new FftConvPlan();
    pl->B = B; pl->B_ir = B_ir; pl->N = N; pl->L = L; pl->nfft = nfft;
    // Route: partitioned overlap-save with LDS-resident 8192-point transforms when the convolution spans at least a few
    // blocks (DDSPP_FFT_PARTITIONED=0: always the whole-signal rocFFT route; =2: always partitioned).
    const int part_opt = ddspp_option_literal("DDSPP_FFT_PARTITIONED", 1);
    if (part_opt == 2 || (part_opt == 1 && (long long)N + L >= 6 * REVERB_PART_BLOCK)) {
        pl->part = true;
        pl->pp.Pn = (L + REVERB_PART_BLOCK - 1) / REVERB_PART_BLOCK;
        pl->pp.nbx = (N + REVERB_PART_BLOCK - 1) / REVERB_PART_BLOCK + 1;
        pl->pp.nbo = (int)(((long long)nfft + REVERB_PART_BLOCK - 1) / REVERB_PART_BLOCK);       // crops stay inside [0, nfft)
        const size_t tbytes = (size_t)2 * REVERB_PART_BLOCK * sizeof(float2);
        float* host = new float[4 * REVERB_PART_BLOCK];
        reverb_part_tables_host(host, host + 2 * REVERB_PART_BLOCK);
        hipError_t e = hipMalloc((void**)&pl->tables, tbytes);
        if (e == hipSuccess) e = hipMemcpy(pl->tables, host, tbytes, hipMemcpyHostToDevice);
        delete[] host;
        if (e != hipSuccess) {
            ddspp_set_error("fftconv_plan_create: twiddle tables: %s", hipGetErrorString(e));
            if (pl->tables) (void)hipFree(pl->tables);
            delete pl;
            return DDSPP_EHIP;
        }
        pl->pp.W = pl->tables;
        pl->pp.U = pl->tables + REVERB_PART_BLOCK;
        size_t off = 0;
        pl->off_xspec = off; off = align256(off + (size_t)B * pl->pp.nbx * REVERB_PART_BLOCK * sizeof(float2));
        pl->off_hspec = off; off = align256(off + (size_t)B_ir * pl->pp.Pn * REVERB_PART_BLOCK * sizeof(float2));
        pl->off_yspec = off; off = align256(off + (size_t)B * pl->pp.nbo * REVERB_PART_BLOCK * sizeof(float2));
        pl->total_bytes = off;
        *out_plan = pl;
        return DDSPP_OK;
    }
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
    const size_t lengths[1] = {(size_t)nfft};
    DDSPP_FFT_CHECK(rocfft_plan_create(&pl->fwd_audio, rocfft_placement_notinplace,
                                       rocfft_transform_type_real_forward, rocfft_precision_single, 1,
                                       lengths, (size_t)B, nullptr));
    DDSPP_FFT_CHECK(rocfft_plan_create(&pl->fwd_ir, rocfft_placement_notinplace,
                                       rocfft_transform_type_real_forward, rocfft_precision_single, 1,
                                       lengths, (size_t)B_ir, nullptr));
    rocfft_plan_description desc = nullptr;
    DDSPP_FFT_CHECK(rocfft_plan_description_create(&desc));
    DDSPP_FFT_CHECK(rocfft_plan_description_set_scale_factor(desc, 1.0 / (double)nfft));
    DDSPP_FFT_CHECK(rocfft_plan_create(&pl->inv, rocfft_placement_notinplace,
                                       rocfft_transform_type_real_inverse, rocfft_precision_single, 1,
                                       lengths, (size_t)B, desc));
    rocfft_plan_description_destroy(desc);
    size_t w0 = 0, w1 = 0, w2 = 0;
    DDSPP_FFT_CHECK(rocfft_plan_get_work_buffer_size(pl->fwd_audio, &w0));
    DDSPP_FFT_CHECK(rocfft_plan_get_work_buffer_size(pl->fwd_ir, &w1));
    DDSPP_FFT_CHECK(rocfft_plan_get_work_buffer_size(pl->inv, &w2));
    pl->rocfft_work_bytes = w0 > w1 ? (w0 > w2 ? w0 : w2) : (w1 > w2 ? w1 : w2);
    DDSPP_FFT_CHECK(rocfft_execution_info_create(&pl->info));
    const size_t nbins = (size_t)nfft / 2 + 1;
    size_t off = 0;
    pl->off_audio_p = off; off = align256(off + (size_t)B * nfft * sizeof(float));
    pl->off_ir_p = off;    off = align256(off + (size_t)B_ir * nfft * sizeof(float));
    pl->off_audio_f = off; off = align256(off + (size_t)B * nbins * sizeof(float2));
    pl->off_ir_f = off;    off = align256(off + (size_t)B_ir * nbins * sizeof(float2));
    pl->off_work = off;    off = align256(off + pl->rocfft_work_bytes);
    pl->total_bytes = off;
    *out_plan = pl;
    return DDSPP_OK;
}

int ddspp_fftconv_plan_destroy(ddspp_fftconv_plan* pl) {
    if (!pl) return DDSPP_OK;
    if (pl->fwd_audio) rocfft_plan_destroy(pl->fwd_audio);
    if (pl->fwd_ir) rocfft_plan_destroy(pl->fwd_ir);
    if (pl->inv) rocfft_plan_destroy(pl->inv);
    if (pl->info) rocfft_execution_info_destroy(pl->info);
    if (pl->tables) (void)hipFree(pl->tables);
    delete pl;
    return DDSPP_OK;
}

size_t ddspp_fftconv_workspace_bytes(const ddspp_fftconv_plan* pl) { return pl ? pl->total_bytes : 0; }
int ddspp_fftconv_fft_size(const ddspp_fftconv_plan* pl) { return pl ? pl->nfft : -1; }

// out[b, n] = (audio[b] * ir'[b])[n + delay] (+ audio[b, n]),  n < out_len
//   ir' = ir with ir'[:, 0] = 0 when mask_dry (ddsp.effects.Reverb._mask_dry_ir)
//   delay = -1 -> (L - 1) // 2 - 1, -2 -> L // 2 (ddsp.core.crop_and_compensate_delay, see ddspp_auto_delay)
// audio rows may be strided (audio_stride >= N floats).  Not re-entrant per plan: one execution of a
// given plan at a time (the rocFFT execution info carries the stream and work buffer).
// First half of ddspp_fftconv_execute: the (dry-masked) impulse responses -> their spectra, kept in `workspace`.
// The impulse response is an input of the call, so a caller can enqueue this early, on another stream, while the
// audio it will convolve is still being synthesised; ddspp_fftconv_execute_prepared then finishes on the same
// workspace once both are done.
int ddspp_fftconv_transform_ir(ddspp_fftconv_plan* pl, const float* ir, int mask_dry, void* workspace,
                               size_t workspace_bytes, hipStream_t stream) {
    DDSPP_REQUIRE(pl && ir && workspace, "fftconv_transform_ir: null argument");
    DDSPP_REQUIRE(workspace_bytes >= pl->total_bytes, "fftconv_transform_ir: workspace too small (%zu < %zu)",
                  workspace_bytes, pl->total_bytes);
    DDSPP_REQUIRE((uintptr_t)workspace % 256 == 0, "fftconv_transform_ir: workspace must be 256-byte aligned");
    char* ws = (char*)workspace;
    if (pl->part)
        return reverb_part_transform_ir(pl->pp, ir, pl->B_ir, pl->L, mask_dry, (float2*)(ws + pl->off_hspec), stream);
    float* ir_p = (float*)(ws + pl->off_ir_p);
    float2* ir_f = (float2*)(ws + pl->off_ir_f);
    const int nfft = pl->nfft;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(stream_grid((size_t)pl->B_ir * (nfft / 4))), dim3(256), 0, stream,
                       ir, ir_p, pl->B_ir, pl->L, pl->L, nfft, mask_dry);
    DDSPP_LAUNCH_CHECK();
    DDSPP_FFT_CHECK(rocfft_execution_info_set_stream(pl->info, stream));
    if (pl->rocfft_work_bytes)
        DDSPP_FFT_CHECK(rocfft_execution_info_set_work_buffer(pl->info, ws + pl->off_work, pl->rocfft_work_bytes));
    void* in2[1] = {ir_p};
    void* out2[1] = {ir_f};
    DDSPP_FFT_CHECK(rocfft_execute(pl->fwd_ir, in2, out2, pl->info));
    return DDSPP_OK;
}

// Second half: audio -> spectrum, product with the impulse-response spectra ddspp_fftconv_transform_ir left in
// `workspace`, inverse transform, crop (+ dry).  Same arguments as ddspp_fftconv_execute, minus the impulse response.
int ddspp_fftconv_execute_prepared(ddspp_fftconv_plan* pl, const float* audio, int audio_stride, float* out,
                                   int out_len, int delay, int add_dry, void* workspace, size_t workspace_bytes,
                                   hipStream_t stream) {
    DDSPP_REQUIRE(pl && audio && out && workspace, "fftconv_execute: null argument");
    DDSPP_REQUIRE(workspace_bytes >= pl->total_bytes, "fftconv_execute: workspace too small (%zu < %zu)",
                  workspace_bytes, pl->total_bytes);
    DDSPP_REQUIRE((uintptr_t)workspace % 256 == 0, "fftconv_execute: workspace must be 256-byte aligned");
    DDSPP_REQUIRE(audio_stride >= pl->N, "fftconv_execute: audio_stride < n_samples");
    const int start = ddspp_auto_delay(delay, pl->L);
    DDSPP_REQUIRE(start >= 0 && out_len > 0 && (long long)start + out_len <= pl->nfft,
                  "fftconv_execute: crop [%d, %d) outside the fft frame of %d", start, start + out_len,
                  pl->nfft);
    DDSPP_REQUIRE(!add_dry || out_len <= pl->N, "fftconv_execute: add_dry needs out_len <= n_samples");
    char* ws = (char*)workspace;
    if (pl->part)
        return reverb_part_execute(pl->pp, audio, audio_stride, pl->B, pl->B_ir, pl->N, (float2*)(ws + pl->off_xspec),
                                   (const float2*)(ws + pl->off_hspec), (float2*)(ws + pl->off_yspec), out, out_len, start,
                                   add_dry, stream);
    float* audio_p = (float*)(ws + pl->off_audio_p);
    float2* audio_f = (float2*)(ws + pl->off_audio_f);
    float2* ir_f = (float2*)(ws + pl->off_ir_f);
    const int nfft = pl->nfft, nbins = nfft / 2 + 1;

    hipLaunchKernelGGL(pad_rows_kernel, dim3(stream_grid((size_t)pl->B * (nfft / 4))), dim3(256), 0, stream,
                       audio, audio_p, pl->B, pl->N, audio_stride, nfft, 0);
    DDSPP_LAUNCH_CHECK();
    DDSPP_FFT_CHECK(rocfft_execution_info_set_stream(pl->info, stream));
    if (pl->rocfft_work_bytes)
        DDSPP_FFT_CHECK(rocfft_execution_info_set_work_buffer(pl->info, ws + pl->off_work, pl->rocfft_work_bytes));
    void* in1[1] = {audio_p};
    void* out1[1] = {audio_f};
    DDSPP_FFT_CHECK(rocfft_execute(pl->fwd_audio, in1, out1, pl->info));
    hipLaunchKernelGGL(spectrum_multiply_kernel, dim3(stream_grid((size_t)pl->B * nbins)), dim3(256), 0, stream,
                       audio_f, ir_f, pl->B, pl->B_ir, nbins);
    DDSPP_LAUNCH_CHECK();
    void* in3[1] = {audio_f};
    void* out3[1] = {audio_p};          // the padded time buffer is free again
    DDSPP_FFT_CHECK(rocfft_execute(pl->inv, in3, out3, pl->info));
    hipLaunchKernelGGL(crop_add_dry_kernel, dim3(stream_grid((size_t)pl->B * out_len)), dim3(256), 0, stream,
                       audio_p, add_dry ? audio : nullptr, out, pl->B, out_len, nfft, start, audio_stride);
    DDSPP_LAUNCH_CHECK();
    return DDSPP_OK;
}

int ddspp_fftconv_execute(ddspp_fftconv_plan* pl, const float* audio, int audio_stride, const float* ir,
                          float* out, int out_len, int delay, int mask_dry, int add_dry, void* workspace,
                          size_t workspace_bytes, hipStream_t stream) {
    DDSPP_REQUIRE(pl && audio && ir && out && workspace, "fftconv_execute: null argument");
    const int rc = ddspp_fftconv_transform_ir(pl, ir, mask_dry, workspace, workspace_bytes, stream);
    if (rc != DDSPP_OK) return rc;
    return ddspp_fftconv_execute_prepared(pl, audio, audio_stride, out, out_len, delay, add_dry, workspace,
                                          workspace_bytes, stream);
}

}  // extern "C"
