#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__global__ void k(float* out) {
    float v = (float)threadIdx.x;
    float s[7];
    v += dpp_take<0xB1, 0xF>(v); s[0] = v;
    v += dpp_take<0x4E, 0xF>(v); s[1] = v;
    v += dpp_take<0x141, 0xF>(v); s[2] = v;
    v += dpp_take<0x140, 0xF>(v); s[3] = v;
    v += dpp_take<0x142, 0xA>(v); s[4] = v;
    v += dpp_take<0x143, 0xC>(v); s[5] = v;
    s[6] = __builtin_amdgcn_readlane(v, 63);
    for (int i = 0; i < 7; ++i) out[i * 64 + threadIdx.x] = s[i];
}
int main() {
    float* d; hipMalloc(&d, 7 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[7 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int i = 0; i < 7; ++i) { printf("step %d:", i); for (int l = 0; l < 64; l += 1) printf(" %g", h[i * 64 + l]); printf("\n"); }
}
