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

template <int CELLS>
__global__ void __launch_bounds__(1024) k_lds_fmax(float* out, int iters) {
  extern __shared__ float ldsf[];
  for (int i = threadIdx.x; i < CELLS; i += blockDim.x) ldsf[i] = 0;
  __syncthreads();
  unsigned s = blockIdx.x * 7919u + threadIdx.x * 104729u + 1;
  for (int i = 0; i < iters; ++i) {
    unsigned r = rng(s);
    __hip_atomic_fetch_max(&ldsf[r % CELLS], (float)(r & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ldsf[0] + ldsf[CELLS - 1];
}

template <int CELLS>
__global__ void __launch_bounds__(1024) k_lds_fadd(float* out, int iters) {
  extern __shared__ float ldsf[];
  for (int i = threadIdx.x; i < CELLS; i += blockDim.x) ldsf[i] = 0;
  __syncthreads();
  unsigned s = blockIdx.x * 7919u + threadIdx.x * 104729u + 1;
  for (int i = 0; i < iters; ++i) {
    unsigned r = rng(s);
    __hip_atomic_fetch_add(&ldsf[r % CELLS], (float)(r & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ldsf[0] + ldsf[CELLS - 1];
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
__global__ void __launch_bounds__(256) k_write_nt(float4* __restrict__ out, size_t n4) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
    { float* o = reinterpret_cast<float*>(&out[i]);
      __builtin_nontemporal_store(1.f, o); __builtin_nontemporal_store(2.f, o + 1);
      __builtin_nontemporal_store(3.f, o + 2); __builtin_nontemporal_store(4.f, o + 3); }
}
// each block owns a contiguous chunk of `per_block` float4
__global__ void __launch_bounds__(256) k_write_chunk(float4* __restrict__ out, size_t n4, int per_block) {
  size_t base = (size_t)blockIdx.x * per_block;
  for (int i = threadIdx.x; i < per_block; i += 256) if (base + i < n4) out[base + i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_copy_chunk(const float4* __restrict__ in, float4* __restrict__ out, size_t n4, int per_block) {
  size_t base = (size_t)blockIdx.x * per_block;
  for (int i = threadIdx.x; i < per_block; i += 256) if (base + i < n4) out[base + i] = in[base + i];
}
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(1024) k_valu_fma(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
    x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
    x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
  }
  float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (r == 123.456f) out[0] = r;
}
__global__ void __launch_bounds__(1024) k_valu_pkfma(float* out, int iters, float a, float b) {
  v2f x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
  v2f va = {a, a}, vb = {b, b};
  for (int i = 0; i < iters; ++i) {
    x0 = __builtin_elementwise_fma(x0, va, vb); x1 = __builtin_elementwise_fma(x1, va, vb);
    x2 = __builtin_elementwise_fma(x2, va, vb); x3 = __builtin_elementwise_fma(x3, va, vb);
  }
  v2f r = x0 + x1 + x2 + x3;
  if (r.x + r.y == 123.456f) out[0] = r.x;
}
// instruction-mix probes: 8 independent chains per wave, iters loop trips
template <int MODE>
__global__ void __launch_bounds__(1024) k_valu_mix(float* out, int iters, float a, float b, int ia) {
  float x[8]; int n[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { x[k] = threadIdx.x + k; n[k] = threadIdx.x * 3 + k; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 0) x[k] = __builtin_fmaf(x[k], a, b);                       // v_fma
      if (MODE == 1) x[k] = x[k] * a;                                          // v_mul
      if (MODE == 2) x[k] = __builtin_floorf(x[k] + b);                        // add + floor
      if (MODE == 3) x[k] = (x[k] > a) ? x[k] - b : x[k] + b;                   // cmp + 2 alu + cndmask
      if (MODE == 4) n[k] = n[k] * ia + 7;                                     // int mad (32-bit mul)
      if (MODE == 5) n[k] = (int)__umul24((unsigned)n[k], (unsigned)ia) + 7;   // mul24 + add
      if (MODE == 6) { n[k] = (int)x[k]; x[k] = (float)(n[k] + 1); }           // cvt i32 <-> f32 + add
      if (MODE == 7) n[k] = (n[k] + ia) ^ (n[k] >> 3);                         // int add/xor/shift
    }
  }
  float r = 0; int m = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { r += x[k]; m += n[k]; }
  if (r == 123.456f || m == 123456789) out[0] = r + m;
}
// ILP probe: CH independent fma chains per wave
template <int CH>
__global__ void __launch_bounds__(1024) k_valu_ilp(float* out, int iters, float a, float b) {
  float x[CH];
#pragma unroll
  for (int k = 0; k < CH; ++k) x[k] = threadIdx.x + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < CH; ++k) x[k] = __builtin_fmaf(x[k], a, b);
  }
  float r = 0;
#pragma unroll
  for (int k = 0; k < CH; ++k) r += x[k];
  if (r == 123.456f) out[0] = r;
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
  {
    const int iters = 4096;
    for (int threads : {256, 512, 1024}) {
      float ms = time_ms([&] { hipLaunchKernelGGL(k_valu_fma, dim3(256), dim3(threads), 0, 0, (float*)dout, iters, 1.0001f, 0.5f); });
      double n = 256.0 * threads * iters * 8;
      printf("v_fma_f32     %4d thr/CU: %.3f ms  %.1f Tlane-instr/s  (%.1f lanes/clk/CU @2.4GHz)\n", threads, ms, n / ms / 1e9, n / ms / 1e-3 / 256 / 2.4e9);
      ms = time_ms([&] { hipLaunchKernelGGL(k_valu_pkfma, dim3(256), dim3(threads), 0, 0, (float*)dout, iters, 1.0001f, 0.5f); });
      n = 256.0 * threads * iters * 4;
      printf("v_pk_fma_f32  %4d thr/CU: %.3f ms  %.1f Tlane-instr/s  (%.1f lanes/clk/CU; x2 flops)\n", threads, ms, n / ms / 1e9, n / ms / 1e-3 / 256 / 2.4e9);
    }
  }
  {
    const int iters = 2048;
    const char* names[8] = {"v_fma_f32 (1 op)", "v_mul_f32 (1 op)", "add+floor (2 ops)", "cmp+sub+add+cndmask (4 ops)",
                            "v_mul_lo_u32+add (2 ops)", "mul_u24+add (2 ops)", "cvt_i32_f32+add+cvt_f32_i32 (3 ops)", "add+shift+xor (3 ops)"};
    const int nops[8] = {1, 1, 2, 4, 2, 2, 3, 3};
    auto run = [&](int mode, auto kern) {
      float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 0, 0, (float*)dout, iters, 1.0001f, 0.5f, 3); });
      double waveinstr = 256.0 * 16 * iters * 8 * nops[mode];
      printf("mix %-40s %.3f ms  %.2f cycles/wave-instr/SIMD @2.1GHz\n", names[mode], ms, ms * 1e-3 * 2.1e9 / (waveinstr / 1024));
    };
    run(0, k_valu_mix<0>); run(1, k_valu_mix<1>); run(2, k_valu_mix<2>); run(3, k_valu_mix<3>);
    run(4, k_valu_mix<4>); run(5, k_valu_mix<5>); run(6, k_valu_mix<6>); run(7, k_valu_mix<7>);
  }
  {
    const int iters = 4096;
    auto run = [&](int ch, int threads, auto kern) {
      float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, (float*)dout, iters, 1.0001f, 0.5f); });
      double waveinstr = 256.0 * (threads / 64) * iters * ch;
      printf("ilp chains=%d waves/SIMD=%d: %.2f cycles/wave-instr/SIMD @2.1GHz\n", ch, threads / 256, ms * 1e-3 * 2.1e9 / (waveinstr / 1024));
    };
    run(1, 1024, k_valu_ilp<1>); run(2, 1024, k_valu_ilp<2>); run(4, 1024, k_valu_ilp<4>); run(8, 1024, k_valu_ilp<8>); run(16, 1024, k_valu_ilp<16>);
    run(1, 512, k_valu_ilp<1>); run(4, 512, k_valu_ilp<4>); run(16, 512, k_valu_ilp<16>);
    run(1, 256, k_valu_ilp<1>); run(4, 256, k_valu_ilp<4>); run(16, 256, k_valu_ilp<16>);
  }
  // LDS atomics: 32K cells (128 KB), 1 block per CU
  {
    const int iters = 2000;
    for (int blocks : {256, 512}) {
      CK(hipFuncSetAttribute((const void*)k_lds_max<32768>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
      float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_max<32768>, dim3(blocks), dim3(1024), 32768 * 4, 0, dout, iters); });
      double n = (double)blocks * 1024 * iters;
      printf("lds_max 128KB  blocks=%d: %.3f ms  %.1f Gatomic/s\n", blocks, ms, n / ms / 1e6);
    }
    CK(hipFuncSetAttribute((const void*)k_lds_fmax<32768>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
    { float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_fmax<32768>, dim3(256), dim3(1024), 32768 * 4, 0, (float*)dout, iters); });
      printf("lds_fmax(f32) 128KB blocks=256: %.3f ms  %.1f Gatomic/s\n", ms, 256.0 * 1024 * iters / ms / 1e6); }
    CK(hipFuncSetAttribute((const void*)k_lds_fadd<32768>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 * 4));
    { float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_fadd<32768>, dim3(256), dim3(1024), 32768 * 4, 0, (float*)dout, iters); });
      printf("lds_fadd(f32) 128KB blocks=256: %.3f ms  %.1f Gatomic/s\n", ms, 256.0 * 1024 * iters / ms / 1e6); }
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
    {
      size_t half = bytes / 2; float4* dst = buf + half / 16;
      for (int blocks : {2048, 8192, 32768}) {
        float ms = time_ms([&] { hipLaunchKernelGGL(k_write_nt, dim3(blocks), dim3(256), 0, 0, buf, bytes / 16); });
        printf("write_nt 1GiB blocks=%d: %.3f ms  %.2f TB/s\n", blocks, ms, bytes / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, buf, dst, half / 16); });
        printf("copy 512MiB->512MiB blocks=%d: %.3f ms  %.2f TB/s (read+write)\n", blocks, ms, bytes / ms / 1e9);
      }
      for (int per_block : {1024, 4096, 16384}) {   // 16 KB, 64 KB, 256 KB chunks
        int blocks = (int)((bytes / 16 + per_block - 1) / per_block);
        float ms = time_ms([&] { hipLaunchKernelGGL(k_write_chunk, dim3(blocks), dim3(256), 0, 0, buf, bytes / 16, per_block); });
        printf("write_chunk 1GiB chunk=%d KB blocks=%d: %.3f ms  %.2f TB/s\n", per_block * 16 / 1024, blocks, ms, bytes / ms / 1e9);
        blocks = (int)((half / 16 + per_block - 1) / per_block);
        ms = time_ms([&] { hipLaunchKernelGGL(k_copy_chunk, dim3(blocks), dim3(256), 0, 0, buf, dst, half / 16, per_block); });
        printf("copy_chunk 512MiB chunk=%d KB blocks=%d: %.3f ms  %.2f TB/s (read+write)\n", per_block * 16 / 1024, blocks, ms, bytes / ms / 1e9);
      }
      // cfg2-sized: write 84 MB
      size_t w84 = (size_t)84 << 20;
      for (int per_block : {1024, 4096}) {
        int blocks = (int)((w84 / 16 + per_block - 1) / per_block);
        float ms = time_ms([&] { hipLaunchKernelGGL(k_write_chunk, dim3(blocks), dim3(256), 0, 0, buf, w84 / 16, per_block); }, 10);
        printf("write_chunk 84MiB chunk=%d KB: %.3f ms  %.2f TB/s\n", per_block * 16 / 1024, ms, w84 / ms / 1e9);
      }
      float ms = time_ms([&] { hipLaunchKernelGGL(k_write_nt, dim3(8192), dim3(256), 0, 0, buf, w84 / 16); }, 10);
      printf("write_nt 84MiB: %.3f ms  %.2f TB/s\n", ms, w84 / ms / 1e9);
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
