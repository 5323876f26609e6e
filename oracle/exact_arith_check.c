/* Test infrastructure: CPU check of the two exact-arithmetic shortcuts the HIP kernels use
 * (ddsp_piano_amd/csrc/ddspp_common.h), against the IEEE operations of the reference:
 *   1. mod_2pi fast path  == floormod(x, float32(2*pi))   (fmodf + sign fix-up = tf.math.floormod)
 *   2. div_const(x, sr)   == x / sr (IEEE)                 for the whitelisted sample rates
 *   3. cos_of_phase_fast's remainder r = s - rint(s/P)*P is exact: equals fmod-based remainder or it - P
 * usage: exact_arith_check [stride]   stride = sample every stride-th float32 (1 = exhaustive, ~6 min)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float mod_fast(float x) {
    const float P = 6.2831855f, INV_UP = 0x1.45f30ap-3f;
    float q = floorf(x * INV_UP);
    float r = fmaf(-q, P, x);
    return r < 0.f ? r + P : r;
}
static inline float mod_ref(float x) {
    const float P = 6.2831855f;
    float t = fmodf(x, P);
    if (t != 0.f && t < 0.f) t += P;
    return t;
}
static inline float rem_rint(float s) {
    const float P = 6.2831855f;
    float q = rintf(s * 0x1.45f306p-3f);
    return fmaf(-q, P, s);
}

int main(int argc, char** argv) {
    uint32_t stride = argc > 1 ? (uint32_t)strtoul(argv[1], 0, 10) : 64;
    if (stride == 0) stride = 1;
    float lim = 2.6e7f;
    uint32_t bl;
    memcpy(&bl, &lim, 4);
    uint64_t bad_mod = 0, bad_rem = 0, n = 0;
    for (uint64_t i = 0; i <= bl; i += stride) {
        uint32_t u = (uint32_t)i;
        float x;
        memcpy(&x, &u, 4);
        float a = mod_fast(x), r = mod_ref(x);
        if (memcmp(&a, &r, 4)) bad_mod++;
        float q = rem_rint(x);
        if (!(q == r || q == r - 6.2831855f)) bad_rem++;     /* r - P is exact (Sterbenz / same grid) */
        n++;
    }
    printf("mod_2pi fast path: %llu mismatches in %llu samples of [0, 2.6e7)\n", (unsigned long long)bad_mod,
           (unsigned long long)n);
    printf("rint remainder   : %llu mismatches\n", (unsigned long long)bad_rem);
    const float rates[] = {8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
    uint64_t bad_div_total = 0;
    for (int k = 0; k < 8; k++) {
        float d = rates[k], rd = 1.0f / d;
        uint64_t bad = 0;
        float xmin = 1e-28f;
        uint32_t b0, b1 = 0x7f000000u;
        memcpy(&b0, &xmin, 4);
        for (uint64_t i = b0; i < b1; i += stride) {
            uint32_t u = (uint32_t)i;
            float x;
            memcpy(&x, &u, 4);
            float q = x * rd;
            float r = fmaf(-q, d, x);
            float y = fmaf(r, rd, q);
            float z = x / d;
            if (memcmp(&y, &z, 4)) bad++;
        }
        printf("div_const d=%g: %llu mismatches for x in [1e-28, 1.7e38)\n", d, (unsigned long long)bad);
        bad_div_total += bad;
    }
    return (bad_mod || bad_rem || bad_div_total) ? 1 : 0;
}
