// Shared host/device helpers for libddspp (gfx950 / CDNA4 only).
//
// Arithmetic contract: every float32 operation that feeds the oscillator
// phase is a separately rounded IEEE op in the order the reference writes it
// (ddsp_piano/modules/inharm_synth.py:49-127 + ddsp.core.angular_cumsum).  The translation units
// are compiled with -ffp-contract=off; fused multiply-adds appear only where they are provably
// exact (div_const, mod_2pi below).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DDSPP_OK 0
#define DDSPP_EINVAL (-22)
#define DDSPP_ENOMEM (-12)
#define DDSPP_EHIP (-5)
#define DDSPP_EFFT (-6)

#define DDSPP_CHUNK 1000          // ddsp.core.angular_cumsum(chunk_size=1000)

// ddsp.core.crop_and_compensate_delay: `start` when delay_compensation < 0.  ddsp 3.7.0 is not on disk, the rule is
// RECALLED: -1 selects `(ir_size - 1) // 2 - 1` (the default; it puts the zero-time tap of the
// cropped 257-tap window, tap 127, at delay 0), -2 selects the alternative recollection `ir_size // 2`.
#define DDSPP_DELAY_AUTO (-1)
#define DDSPP_DELAY_AUTO_HALF (-2)
static inline int ddspp_auto_delay(int delay_compensation, int ir_size) {
    if (delay_compensation >= 0) return delay_compensation;
    return delay_compensation == DDSPP_DELAY_AUTO_HALF ? ir_size / 2 : (ir_size - 1) / 2 - 1;
}
#define DDSPP_WAVE 64

// ddspp_fdn_transfer(solve): how the D x D system of a frequency bin is solved (fdn.hip)
#define DDSPP_FDN_SOLVE_F64 0
#define DDSPP_FDN_SOLVE_C64_INVERSE 1

extern "C" void ddspp_set_error(const char* fmt, ...);
// tuning option `name` (a DDSPP_* environment variable, read once and cached; ddspp_set_option / ddspp_reload_options)
extern "C" int ddspp_option(const char* name, int dflt);
// the library's own call sites: `name` must be a string LITERAL (its address is the cache key) -- no lock, no allocation
extern "C" __attribute__((visibility("hidden"))) int ddspp_option_literal(const char* name, int dflt);

#define DDSPP_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ddspp_set_error(__VA_ARGS__);        \
            return DDSPP_EINVAL;                 \
        }                                        \
    } while (0)

#define DDSPP_HIP_CHECK(expr)                                                          \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            ddspp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                            __FILE__, __LINE__);                                       \
            return DDSPP_EHIP;                                                         \
        }                                                                              \
    } while (0)

#define DDSPP_LAUNCH_CHECK()                                                           \
    do {                                                                               \
        hipError_t _e = hipGetLastError();                                             \
        if (_e != hipSuccess) {                                                        \
            ddspp_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                       \
            return DDSPP_EHIP;                                                         \
        }                                                                              \
    } while (0)

// float32(2*pi) = 6.2831855f : what TensorFlow makes of the python float `2.0 * pi`
// (inharm_synth.py:69) and the modulus of ddsp.core.angular_cumsum.
#define DDSPP_TWO_PI_F32 6.2831855f

#ifdef __HIPCC__
namespace ddspp {

// ------------------------------------------------------------------------------------------
// scale functions of the processors' get_controls (ddsp.core.exp_sigmoid, inharm_synth.py:13-17 exp_tanh)
// ------------------------------------------------------------------------------------------
enum { SCALE_NONE = 0, SCALE_EXP_SIGMOID = 1, SCALE_EXP_TANH = 2 };

struct ScaleFn {
    int kind;
    float log_exponent;   // float32(log(exponent))
    float max_value;
    float threshold;
    float gain;           // exp_tanh only
};

// exp / log through the hardware v_exp_f32 / v_log_f32 (base 2, ~1 ulp): the scale functions shape
// amplitudes, never phases, and TF's own sigmoid / pow are a few ulp from correctly rounded anyway.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504f); }
__device__ __forceinline__ float fast_pow(float b, float e) {      // b >= 0
    return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(b));
}

// sigmoid(y) ** p = exp(-p log(1 + exp(-y))): three transcendentals (exp2, log2, exp2) instead of the four of
// rcp + pow (round 3: a quarter of the get_controls kernel's issue slots were these), and one rounding less.
__device__ __forceinline__ float sigmoid_pow(float y, float p) {
    return __builtin_amdgcn_exp2f(-p * __builtin_amdgcn_logf(1.0f + fast_exp(-y)));
}

// The same with the kind a compile-time constant, and a dispatcher that decides it ONCE around a loop: apply_scale's
// run-time `kind` costs a maze of scalar branches per ELEMENT when the compiler cannot hoist it (round 4: eight branches
// per magnitude in the FilteredNoise kernels' staging, 2 us per unit of 32 frames; 76 instead of 28 instructions per
// harmonic in get_controls).
template <int KIND>
__device__ __forceinline__ float scale_of(const ScaleFn& s, float x) {
    if (KIND == SCALE_EXP_SIGMOID) return s.max_value * sigmoid_pow(x, s.log_exponent) + s.threshold;
    if (KIND == SCALE_EXP_TANH) return s.max_value * sigmoid_pow(2.0f * (s.gain * x), s.log_exponent) + s.threshold;
    return x;
}
template <int KIND>
struct ScaleKind {
    static constexpr int value = KIND;
};
// f(ScaleKind<k>{}) with k = the kind, or -1 for "leave the values alone" (kind < 0)
template <class F>
__device__ __forceinline__ void with_scale_kind(int kind, F&& f) {
    if (kind == SCALE_EXP_SIGMOID) f(ScaleKind<SCALE_EXP_SIGMOID>{});
    else if (kind == SCALE_EXP_TANH) f(ScaleKind<SCALE_EXP_TANH>{});
    else if (kind == SCALE_NONE) f(ScaleKind<SCALE_NONE>{});
    else f(ScaleKind<-1>{});
}
// raw magnitudes -> scale_fn(m + bias), four at a time (kind -1: untouched)
template <int KIND>
__device__ __forceinline__ float4 scale4_of(const ScaleFn& s, float4 m, float bias) {
    if (KIND < 0) return m;
    return make_float4(scale_of<KIND>(s, m.x + bias), scale_of<KIND>(s, m.y + bias), scale_of<KIND>(s, m.z + bias),
                       scale_of<KIND>(s, m.w + bias));
}

__device__ __forceinline__ float apply_scale(const ScaleFn& s, float x) {
    if (s.kind == SCALE_EXP_SIGMOID) {
        // max_value * sigmoid(x) ** log(exponent) + threshold          (ddsp.core.exp_sigmoid)
        return s.max_value * sigmoid_pow(x, s.log_exponent) + s.threshold;
    }
    if (s.kind == SCALE_EXP_TANH) {
        // max_value * (0.5 * (tanh(gain * x) + 1)) ** log(exponent) + threshold   (inharm_synth.py:13-17)
        // 0.5 * (tanh(y) + 1) == sigmoid(2 y)
        return s.max_value * sigmoid_pow(2.0f * (s.gain * x), s.log_exponent) + s.threshold;
    }
    return x;
}



// ------------------------------------------------------------------------------------------
// Correctly rounded x / d for a constant d, given rd = RN(1/d)  (Markstein's correction step).
//   q  = RN(x * rd)           faithful quotient
//   r  = x - q * d            exact in one FMA
//   q' = RN(q + r * rd)       = RN(x / d)
// Valid away from under/overflow; the caller enables it only for sample rates whose result was
// checked exhaustively against IEEE division (tests/test_exact_arith.py) and falls back to the
// IEEE divide otherwise.  Replaces `omegas / float(sample_rate)` (inharm_synth.py:70).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float div_const(float x, float d, float rd) {
    float q = x * rd;
    float r = __builtin_fmaf(-q, d, x);
    return __builtin_fmaf(r, rd, q);
}

// ------------------------------------------------------------------------------------------
// floormod(x, float32(2*pi)) exactly as tf.math.floormod / np.mod compute it
// (fmod, then `+ y` when the remainder is non-zero and negative).
// Fast path, 0 <= x < 2^22 * 2pi:  q = floor(x * INV) with INV rounded UP by 2 ulp, so q is
// floor(x / P) or one more, never less; r = x - q * P is then exact in one FMA because both x
// and q * P are multiples of ulp(P) = 2^-21 and |r| < 8; a negative r gets + P (exact).
// ------------------------------------------------------------------------------------------
__device__ __attribute__((noinline)) float mod_2pi_slow(float x) {
    const float P = DDSPP_TWO_PI_F32;
    float t = fmodf(x, P);
    if (t != 0.0f && t < 0.0f) t = t + P;
    return t;
}

__device__ __forceinline__ float mod_2pi(float x) {
    const float P = DDSPP_TWO_PI_F32;
    const float INV_UP = 0x1.45f30ap-3f;   // RN(1/P) = 0x1.45f306p-3 rounded up by two more ulps
    if (__builtin_expect(!(x >= 0.0f && x < 2.6e7f), 0)) return mod_2pi_slow(x);
    float q = __builtin_floorf(x * INV_UP);
    float r = __builtin_fmaf(-q, P, x);
    return r < 0.0f ? r + P : r;
}

// cos of a phase already reduced to [0, 2pi]: hardware v_cos_f32 takes revolutions.
__device__ __forceinline__ float cos_reduced(float r) {
    return __builtin_amdgcn_cosf(r * 0x1.45f306p-3f);
}

// cos(floormod(s, P)) for 0 <= s < 2^22 * P without materialising the floormod:
//   q = rint(s / P), r = s - q * P  (exact, one FMA: multiples of 2^-21 below 8 in magnitude)
// r is floormod(s, P) or floormod(s, P) - P; cos is evaluated at r, i.e. at most P - 2pi = 1.75e-7
// rad away from the reference's argument -- below float32 resolution of the cosine itself.
__device__ __forceinline__ float cos_of_phase_fast(float s) {
    const float P = DDSPP_TWO_PI_F32;
    const float q = __builtin_rintf(s * 0x1.45f306p-3f);
    const float r = __builtin_fmaf(-q, P, s);
    return __builtin_amdgcn_cosf(r * 0x1.45f306p-3f);
}

__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

}  // namespace ddspp
#endif
