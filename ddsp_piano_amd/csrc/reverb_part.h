// Partitioned overlap-save convolution (csrc/reverb_part.hip), driven by the plan object of reverb.hip.
#pragma once
#include "ddspp_common.h"

#define REVERB_PART_BLOCK 4096      // hop in real samples = complex transform size (16^3)

namespace ddspp {

struct PartPlan {
    int Pn;              // impulse-response partitions: ceil(L / block)
    int nbx;             // audio windows a workspace holds: ceil(N / block) + 1
    int nbo;             // output blocks a workspace holds: ceil((N + L - 1) / block)
    const float2* W;     // device: exp(-2 pi i e / 4096), e < 4096
    const float2* U;     // device: exp(-2 pi i k / 8192), k < 4096
};

// host: the two twiddle tables, 2 * 4096 floats each (double precision sin / cos, rounded once)
void reverb_part_tables_host(float* W, float* U);
// impulse responses [B_ir, L] -> spectra Hspec [B_ir, Pn, 4096] (complex; ir[:, 0] zeroed when mask_dry)
int reverb_part_transform_ir(const PartPlan& pp, const float* ir, int B_ir, int L, int mask_dry, float2* Hspec,
                             hipStream_t stream);
// audio [B, N] (row stride audio_stride) -> Xspec [B, <= nbx, 4096] -> Yspec [B, <= nbo, 4096];
// out[b, n] = (audio * ir)[n + start] (+ audio[b, n])
int reverb_part_execute(const PartPlan& pp, const float* audio, int audio_stride, int B, int B_ir, int N, float2* Xspec,
                        const float2* Hspec, float2* Yspec, float* out, int out_len, int start, int add_dry, hipStream_t stream);

}  // namespace ddspp
