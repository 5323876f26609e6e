// How many 256-thread workgroups per CU does the runtime place for a given dynamic LDS size?  (allocation granule of gfx950)
// Measured two ways: hipOccupancyMaxActiveBlocksPerMultiprocessor, and a kernel whose workgroups spin until a flag says
// that all of them are resident -- the largest grid that does not dead-lock is the number of slots.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(256) touch(float* out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[255];
}

__global__ void __launch_bounds__(256) resident(int* counter, int want, int* ok, long long budget) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(counter, 1);
        long long t0 = wall_clock64();
        while (atomicAdd(counter, 0) < want && wall_clock64() - t0 < budget) __builtin_amdgcn_s_sleep(8);
        if (atomicAdd(counter, 0) >= want) atomicAdd(ok, 1);
    }
    __syncthreads();
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    printf("CUs %d, sharedMemPerMultiprocessor %zu, maxSharedMemoryPerBlock %zu\n", prop.multiProcessorCount,
           (size_t)prop.maxSharedMemoryPerMultiProcessor, (size_t)prop.sharedMemPerBlock);
    int *counter, *ok;
    hipMalloc(&counter, 4);
    hipMalloc(&ok, 4);
    const int cus = prop.multiProcessorCount;
    for (int lds = 51200; lds <= 55296; lds += 256) {
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, touch, 256, lds);
        int best = 0;
        for (int per = 1; per <= 4; ++per) {
            hipMemset(counter, 0, 4);
            hipMemset(ok, 0, 4);
            hipLaunchKernelGGL(resident, dim3(cus * per), dim3(256), lds, 0, counter, cus * per, ok, 20000000LL);
            hipDeviceSynchronize();
            int h = 0;
            hipMemcpy(&h, ok, 4, hipMemcpyDeviceToHost);
            if (h == cus * per) best = per;
        }
        printf("lds %6d B: occupancy API %d blocks/CU, co-resident test %d blocks/CU\n", lds, nb, best);
    }
    return 0;
}
