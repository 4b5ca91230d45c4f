// Speed of light of the cfg2 launch's traffic on this chip: the same grid (strips x frames, 1024 threads,
// one workgroup per CU), the same bytes (each workgroup reads its 160-column strip of a 640x480 depth map
// with 16-byte loads, four rows per thread in flight twice over, and writes its share of a 512x512 map
// + mask), no geometry, no LDS window, no arithmetic -- on ROT rotating buffer sets (> 256 MiB live).
//   hipcc --offload-arch=gfx950 -O3 tools/skeleton.hip -o tools/tmp/skeleton && tools/tmp/skeleton
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ROTATE: frame b starts its image rows at row (b * 37) % H and its fill rows at an offset of (b * 5) steps, wrapping
// around -- do 64 workgroups walking 64 equally laid out frames in lockstep camp on the same memory channels?
template <bool NT, bool READ, bool WRITE, bool ROTATE = false>
__global__ void __launch_bounds__(1024) k_skel(const float* depth, float* out, unsigned char* mask, float* sink,
                                               int P, int H, int W, int mh, int mw) {
  const int part = blockIdx.x, b = blockIdx.z;
  const int rot_r = ROTATE ? (b * 37) % H : 0;
  const int wp = W / P, nx = wp / 4, rows_per_iter = 1024 / nx;
  const int gx = threadIdx.x % nx, gy = threadIdx.x / nx;
  const float* img = depth + (size_t)b * H * W + part * wp + gx * 4;
  f32x4 acc = {0, 0, 0, 0};
  const int step = rows_per_iter * 4;
  f32x4 za[4], zb[4];
  auto load = [&](f32x4 (&z)[4], int r) {
#pragma unroll
    for (int u = 0; u < 4; ++u) { int rr = r + u * rows_per_iter; rr = rr < H ? rr : H - 1; rr += rot_r; rr = rr >= H ? rr - H : rr; z[u] = *reinterpret_cast<const f32x4*>(img + (size_t)rr * W); }
  };
  // fill rows part, part + P, ...: wave v rows part + (v + 16 j) P, 256 cells per step
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* omap = out + (size_t)b * mh * mw;
  unsigned char* mmap = mask + (size_t)b * mh * mw;
  const int rows_mine = (mh - part + P - 1) / P;
  const int chunks = (mw + 255) >> 8;
  const int fill_steps = WRITE && wave < rows_mine ? ((rows_mine - wave + 15) >> 4) * chunks : 0;
  int fs = 0, f_row = part + wave * P, f_chunk = 0;
  const int rot_f = ROTATE ? (b * 5 * 16 * P) % mh : 0;
  auto fill_step = [&]() {
    if (fs < fill_steps) {
      const int x = (f_chunk << 8) + (lane << 2);
      if (x < mw) {
        const f32x4 v = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int fr = f_row + rot_f; fr = fr >= mh ? fr - mh : fr;
        f32x4* p = reinterpret_cast<f32x4*>(omap + (size_t)fr * mw + x);
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        *reinterpret_cast<unsigned*>(mmap + (size_t)fr * mw + x) = 0u;
      }
      ++fs;
      const bool next = f_chunk + 1 == chunks;
      f_chunk = next ? 0 : f_chunk + 1;
      f_row += next ? 16 * P : 0;
    }
  };
  if (READ) {
    int r = gy;
    load(za, r); load(zb, r + step);
    const int niter = (H + step - 1) / step;
    for (int it = 0; it < niter; it += 2) {
      fill_step(); fill_step();
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += za[u];
      if (it + 1 < niter) {
        load(za, r + 2 * step);
        fill_step(); fill_step();
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += zb[u];
        load(zb, r + 3 * step);
      }
      r += 2 * step;
    }
  }
  while (fs < fill_steps) fill_step();
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

int main() {
  const int B = 64, H = 480, W = 640, mh = 512, mw = 512, P = 4, ROT = 5;
  std::vector<float*> d(ROT), o(ROT); std::vector<unsigned char*> m(ROT);
  float* sink;
  CK(hipMalloc(&sink, 256));
  for (int i = 0; i < ROT; ++i) {
    CK(hipMalloc(&d[i], (size_t)B * H * W * 4)); CK(hipMemset(d[i], 0x3f, (size_t)B * H * W * 4));
    CK(hipMalloc(&o[i], (size_t)B * mh * mw * 4)); CK(hipMalloc(&m[i], (size_t)B * mh * mw));
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kern, int rot, int Pk) {
    for (int rep = 0; rep < 3; ++rep) {
      for (int j = 0; j < 8; ++j) hipLaunchKernelGGL(kern, dim3(Pk, 1, B), dim3(1024), 0, 0, d[j % rot], o[j % rot], m[j % rot], sink, Pk, H, W, mh, mw);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      const int n = 64;
      for (int j = 0; j < n; ++j) hipLaunchKernelGGL(kern, dim3(Pk, 1, B), dim3(1024), 0, 0, d[j % rot], o[j % rot], m[j % rot], sink, Pk, H, W, mh, mw);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / n;
      printf("%-34s rot=%d P=%d: %6.2f us per launch   %5.0f GB/s of 162.5 MB\n", name, rot, Pk, us, 162.52928 / us * 1e3);
    }
  };
  run("read+write nt", k_skel<true, true, true>, ROT, P);
  run("read+write default policy", k_skel<false, true, true>, ROT, P);
  run("read only", k_skel<true, true, false>, ROT, P);
  run("write only nt", k_skel<true, false, true>, ROT, P);
  run("read+write nt", k_skel<true, true, true>, 1, P);
  run("read+write nt (8 strips: 2 rounds)", k_skel<true, true, true>, ROT, 8);
  run("read+write nt, rows rotated per frame", k_skel<true, true, true, true>, ROT, P);
  run("read only, rows rotated per frame", k_skel<true, true, false, true>, ROT, P);
  run("write only nt, rows rotated per frame", k_skel<true, false, true, true>, ROT, P);
  return 0;
}
