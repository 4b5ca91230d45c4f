// Micro-benchmarks that size the projector's design choices on MI355X:
// LDS atomic-max rate, global atomic-max rate (agent vs workgroup scope),
// streaming read/write bandwidth.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline unsigned rng(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int CELLS>
__global__ void __launch_bounds__(1024) k_lds_max(unsigned* out, int iters) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < CELLS; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  unsigned s = blockIdx.x * 7919u + threadIdx.x * 104729u + 1;
  for (int i = 0; i < iters; ++i) {
    unsigned r = rng(s);
    atomicMax(&lds[r % CELLS], r);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + lds[CELLS - 1];
}

template <int SCOPE>
__global__ void __launch_bounds__(256) k_glb_max(unsigned* canvas, unsigned cells_per_region, int regions, int iters) {
  unsigned s = blockIdx.x * 7919u + threadIdx.x * 104729u + 1;
  unsigned* base = canvas + (size_t)(blockIdx.x % regions) * cells_per_region;
  for (int i = 0; i < iters; ++i) {
    unsigned r = rng(s);
    __hip_atomic_fetch_max(base + (r % cells_per_region), r, __ATOMIC_RELAXED, SCOPE);
  }
}

__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ in, float* out, size_t n4) {
  float acc = 0;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = in[i]; acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(float4* __restrict__ out, size_t n4) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_empty() {}

template <class F> float time_ms(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz L2 %d\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.l2CacheSize);
  unsigned* dout; CK(hipMalloc(&dout, 1 << 20));
  // LDS atomics: 32K cells (128 KB), 1 block per CU
  {
    const int iters = 2000;
    for (int blocks : {256, 512}) {
      CK(hipFuncSetAttribute((const void*)k_lds_max<32768>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
      float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_max<32768>, dim3(blocks), dim3(1024), 32768 * 4, 0, dout, iters); });
      double n = (double)blocks * 1024 * iters;
      printf("lds_max 128KB  blocks=%d: %.3f ms  %.1f Gatomic/s\n", blocks, ms, n / ms / 1e6);
    }
    CK(hipFuncSetAttribute((const void*)k_lds_max<8192>, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 4));
    float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_max<8192>, dim3(1024), dim3(1024), 8192 * 4, 0, dout, iters); });
    printf("lds_max 32KB   blocks=1024: %.3f ms  %.1f Gatomic/s\n", ms, 1024.0 * 1024 * iters / ms / 1e6);
  }
  // global atomics on 64 regions of 1 MB (256K cells)
  {
    unsigned* canvas; size_t cells = 262144; int regions = 64;
    CK(hipMalloc(&canvas, cells * regions * 4)); CK(hipMemset(canvas, 0, cells * regions * 4));
    const int iters = 200; int blocks = 2048;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_glb_max<__HIP_MEMORY_SCOPE_AGENT>, dim3(blocks), dim3(256), 0, 0, canvas, (unsigned)cells, regions, iters); });
    printf("global_max agent scope random: %.3f ms  %.2f Gatomic/s\n", ms, (double)blocks * 256 * iters / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_glb_max<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(blocks), dim3(256), 0, 0, canvas, (unsigned)cells, regions, iters); });
    printf("global_max workgroup scope random (64 regions): %.3f ms  %.2f Gatomic/s\n", ms, (double)blocks * 256 * iters / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_glb_max<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(blocks), dim3(256), 0, 0, canvas, (unsigned)cells, 8, iters); });
    printf("global_max workgroup scope random (8 regions, region = block%%8): %.3f ms  %.2f Gatomic/s\n", ms, (double)blocks * 256 * iters / ms / 1e6);
    CK(hipFree(canvas));
  }
  // streaming
  {
    size_t bytes = (size_t)1 << 30; float4* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    for (int blocks : {2048, 8192}) {
      float ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, buf, (float*)dout, bytes / 16); });
      printf("read  1GiB blocks=%d: %.3f ms  %.2f TB/s\n", blocks, ms, bytes / ms / 1e9);
      ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, buf, bytes / 16); });
      printf("write 1GiB blocks=%d: %.3f ms  %.2f TB/s\n", blocks, ms, bytes / ms / 1e9);
    }
    // L3-resident size: 160 MB (cfg2 working set)
    size_t small = (size_t)80 << 20;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, buf, (float*)dout, small / 16); }, 10);
    printf("read  80MiB (L3-resident re-read): %.3f ms  %.2f TB/s\n", ms, small / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, 0, buf, small / 16); }, 10);
    printf("write 80MiB: %.3f ms  %.2f TB/s\n", ms, small / ms / 1e9);
    CK(hipFree(buf));
  }
  {
    float ms = time_ms([&] { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0); });
    printf("empty kernel: %.2f us per launch (100 back-to-back)\n", ms * 10);
  }
  return 0;
}
