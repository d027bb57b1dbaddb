// Does the L2 keep rewritten lines, or do rewrites reach HBM?  One workgroup rewrites a small buffer many times with different
// store widths; WRITE_SIZE per kernel (rocprofv3 --pmc WRITE_SIZE) against the bytes stored tells.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REPS 64
__global__ void k_x4(uint4 *p, int n16) { for (int r = 0; r < REPS; r++) { for (int i = threadIdx.x; i < n16; i += blockDim.x) p[i] = make_uint4(r, i, r, i); __syncthreads(); } }
__global__ void k_b32(uint32_t *p, int n4) { for (int r = 0; r < REPS; r++) { for (int i = threadIdx.x; i < n4; i += blockDim.x) p[i] = r + i; __syncthreads(); } }
__global__ void k_b16(uint16_t *p, int n2) { for (int r = 0; r < REPS; r++) { for (int i = threadIdx.x; i < n2; i += blockDim.x) p[i] = (uint16_t)(r + i); __syncthreads(); } }
__global__ void k_b8(uint8_t *p, int n1) { for (int r = 0; r < REPS; r++) { for (int i = threadIdx.x; i < n1; i += blockDim.x) p[i] = (uint8_t)(r + i); __syncthreads(); } }
// sparse: only the first 64 B of every 256-B slot (the token slots' pattern)
__global__ void k_sparse(uint32_t *p, int nslots) { for (int r = 0; r < REPS; r++) { for (int i = threadIdx.x; i < nslots * 16; i += blockDim.x) p[(i >> 4) * 64 + (i & 15)] = r + i; __syncthreads(); } }
// write, then another wave reads it back (the consumer pattern), then rewrite
__global__ void k_rw(uint32_t *p, int n4, uint32_t *out) { uint32_t acc = 0; for (int r = 0; r < REPS; r++) { for (int i = threadIdx.x; i < n4; i += blockDim.x) p[i] = r + i; __syncthreads();
    for (int i = (threadIdx.x + 64) % blockDim.x; i < n4; i += blockDim.x) acc += p[i]; __syncthreads(); } out[threadIdx.x] = acc; }
// private (scratch) memory rewritten: dynamic indexing keeps the array out of registers
__global__ void k_scratch(uint32_t *out, int n) { uint32_t a[128]; for (int i = 0; i < 128; i++) a[i] = i * threadIdx.x; uint32_t acc = 0;
    for (int r = 0; r < REPS * 16; r++) { int k = (r * 37 + threadIdx.x + n) & 127; a[k] += acc + r; acc ^= a[(k * 5 + 3) & 127]; } out[threadIdx.x] = acc; }
// plain stores, then agent-scope (sc1) loads of the same lines by another wave
__global__ void k_sc1load(uint32_t *p, int n4, uint32_t *out) { uint32_t acc = 0; for (int r = 0; r < REPS; r++) { for (int i = threadIdx.x; i < n4; i += blockDim.x) p[i] = r + i; __syncthreads();
    for (int i = (threadIdx.x + 64) % blockDim.x; i < n4; i += blockDim.x) acc += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __syncthreads(); } out[threadIdx.x] = acc; }
// agent-scope relaxed stores (sc1)
__global__ void k_sc1store(uint32_t *p, int n4) { for (int r = 0; r < REPS; r++) { for (int i = threadIdx.x; i < n4; i += blockDim.x) __hip_atomic_store(p + i, r + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __syncthreads(); } }
// plain stores with an agent-scope release fence per pass (buffer_wbl2)
__global__ void k_relfence(uint32_t *p, int n4) { for (int r = 0; r < REPS; r++) { for (int i = threadIdx.x; i < n4; i += blockDim.x) p[i] = r + i; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __syncthreads(); } }
// scalar (s_load) reads of lines written by vector stores
__global__ void k_sload(uint32_t *p, int n4, uint32_t *out) { uint32_t acc = 0; for (int r = 0; r < REPS; r++) { for (int i = threadIdx.x; i < n4; i += blockDim.x) p[i] = r + i; __syncthreads();
    const uint32_t *q = (const uint32_t *)__builtin_assume_aligned(p, 16); for (int i = 0; i < n4; i += 64) acc += __builtin_nontemporal_load(q + __builtin_amdgcn_readfirstlane(i)); __syncthreads(); } out[threadIdx.x] = acc; }
// the same 1 MiB rewritten 2000 times over ~0.5 s (pauses between passes): does the L2 clean dirty lines in the background?
__global__ void k_slow(uint32_t *p, int n4, int passes) { for (int r = 0; r < passes; r++) { for (int i = threadIdx.x; i < n4; i += blockDim.x) p[i] = r + i; __syncthreads();
    long long t0 = wall_clock64(); while (wall_clock64() - t0 < 20000) __builtin_amdgcn_s_sleep(32); __syncthreads(); } }
// footprint sweep: 32 rewrites of n16 * 16 bytes by one workgroup -- where does the L2 stop holding them?
__global__ void k_foot(uint4 *p, int n16) { for (int r = 0; r < 32; r++) { for (int i = threadIdx.x; i < n16; i += blockDim.x) p[i] = make_uint4(r, i, r, i); __syncthreads(); } }
int main() {
    const size_t bytes = 1 << 20; uint8_t *d; uint32_t *o; hipMalloc(&d, bytes); hipMalloc(&o, 4096); hipMemset(d, 0, bytes);
    hipLaunchKernelGGL(k_x4, 1, 512, 0, 0, (uint4 *)d, (int)(bytes / 16));
    hipLaunchKernelGGL(k_b32, 1, 512, 0, 0, (uint32_t *)d, (int)(bytes / 4));
    hipLaunchKernelGGL(k_b16, 1, 512, 0, 0, (uint16_t *)d, (int)(bytes / 8));
    hipLaunchKernelGGL(k_b8, 1, 512, 0, 0, d, (int)(bytes / 16));
    hipLaunchKernelGGL(k_sparse, 1, 512, 0, 0, (uint32_t *)d, (int)(bytes / 256));
    hipLaunchKernelGGL(k_rw, 1, 512, 0, 0, (uint32_t *)d, (int)(bytes / 4), o);
    hipLaunchKernelGGL(k_scratch, 1, 512, 0, 0, o, 3);
    hipLaunchKernelGGL(k_sc1load, 1, 512, 0, 0, (uint32_t *)d, (int)(bytes / 4), o);
    hipLaunchKernelGGL(k_sc1store, 1, 512, 0, 0, (uint32_t *)d, (int)(bytes / 4));
    hipLaunchKernelGGL(k_relfence, 1, 512, 0, 0, (uint32_t *)d, (int)(bytes / 4));
    hipLaunchKernelGGL(k_sload, 1, 512, 0, 0, (uint32_t *)d, (int)(bytes / 4), o);
    hipLaunchKernelGGL(k_slow, 1, 512, 0, 0, (uint32_t *)d, (int)(bytes / 4), 2000);
    { uint8_t *big; hipMalloc(&big, 16 << 20); hipMemset(big, 0, 16 << 20); hipDeviceSynchronize();
      const int kb[] = {512, 1024, 1536, 2048, 3072, 4096, 6144, 8192};
      for (int k : kb) hipLaunchKernelGGL(k_foot, 1, 512, 0, 0, (uint4 *)big, k * 1024 / 16); }
    hipDeviceSynchronize(); printf("done: %d rewrites of a buffer; x4/b32: 1 MiB, b16: 256 KiB, b8: 64 KiB, sparse: 4096 slots x 64 B, rw: 1 MiB\n", REPS); return 0;
}
