#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512, 2) probe(unsigned *out) {
    extern __shared__ unsigned char lds[];
    lds[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id + (lds[threadIdx.x] & 0);
}
int main() {
    unsigned *d, h[8 * 4];
    (void)hipMalloc(&d, sizeof h);
    (void)hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 156832);
    hipLaunchKernelGGL(probe, dim3(4), dim3(512), 156832, 0, d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; b++) { printf("block %d:", b); for (int w = 0; w < 8; w++) printf("  w%d simd=%u wave=%u cu=%u", w, (h[b*8+w] >> 4) & 3, h[b*8+w] & 15, (h[b*8+w] >> 8) & 15); printf("\n"); }
    return 0;
}
