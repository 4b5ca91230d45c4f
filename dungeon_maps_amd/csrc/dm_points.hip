// Point-set primitives with the reference's exact float32 semantics, for callers
// that work on materialised tensors instead of the fused projector: map fusion
// (fuse_topdown_maps), the public project()/scatter_tensor(), map_quantize() and
// the space transforms on GPU tensors.  Low volume, latency-insensitive: simple
// one-thread-per-element kernels, global atomics for the scatter.
//
//   k_affine_points   utils.rotate + utils.translate (utils.py:229-330):
//                     out_i = fma(p2,R[6+i], fma(p1,R[3+i], p0*R[i])) (+ t_i),
//                     translate before or after the rotation
//   k_map_quantize    maps.py:944-1019 (true division, round half up, int64)
//   k_scatter_flat    utils.scatter_tensor incl. the torch_scatter call
//                     (utils.py:389-492) on pre-ravelled indices
#include <stdlib.h>
#include <string.h>

#include "dm_kernels.hpp"
#include "dm_window_geometry.hpp"      // exact_reciprocal

namespace dm {

namespace {

__global__ void __launch_bounds__(256)
k_affine_points(const float* __restrict__ pts, const float* __restrict__ R,
                const float* __restrict__ t, int translate_first, size_t n_per_batch,
                float* __restrict__ out) {
  const int b = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_per_batch) return;
  const float* r = R + 9 * b;
  const float* tb = t + 3 * b;
  const float* p = pts + ((size_t)b * n_per_batch + i) * 3;
  float p0 = p[0], p1 = p[1], p2 = p[2];
  if (translate_first) { p0 += tb[0]; p1 += tb[1]; p2 += tb[2]; }
  float o0 = __builtin_fmaf(p2, r[6], __builtin_fmaf(p1, r[3], p0 * r[0]));
  float o1 = __builtin_fmaf(p2, r[7], __builtin_fmaf(p1, r[4], p0 * r[1]));
  float o2 = __builtin_fmaf(p2, r[8], __builtin_fmaf(p1, r[5], p0 * r[2]));
  if (!translate_first) { o0 += tb[0]; o1 += tb[1]; o2 += tb[2]; }
  float* o = out + ((size_t)b * n_per_batch + i) * 3;
  o[0] = o0; o[1] = o1; o[2] = o2;
}

// float -> int64 the way x86 does it for the reference's `.to(torch.int64)`:
// NaN and out-of-range become INT64_MIN
__device__ inline long long to_i64_x86(float f) {
  if (!(f >= -9.2233720368547758e18f && f < 9.2233720368547758e18f)) return (long long)0x8000000000000000ull;
  return (long long)f;
}

__global__ void __launch_bounds__(256)
k_map_quantize(const float* __restrict__ x, const float* __restrict__ z,
               const float* __restrict__ woff, const float* __restrict__ hoff, float res,
               float mhm1, int flip, size_t n_per_batch, long long* __restrict__ xb,
               long long* __restrict__ zb) {
  const int b = blockIdx.y;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_per_batch) return;
  const size_t k = (size_t)b * n_per_batch + i;
  float xf = x[k] / res + woff[b];
  float zf = z[k] / res + hoff[b];
  if (flip) zf = mhm1 - zf;
  xb[k] = to_i64_x86(__builtin_floorf(xf + 0.5f));
  zb[k] = to_i64_x86(__builtin_floorf(zf + 0.5f));
}

__device__ inline void atomic_max_f(float* addr, float v) {
  if (!(v == v)) return;
  v += 0.0f;
  if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ inline void atomic_min_f(float* addr, float v) {
  if (!(v == v)) return;
  v += 0.0f;
  if (v >= 0.0f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ inline void atomic_mul_f(float* addr, float v) {
  unsigned int* a = reinterpret_cast<unsigned int*>(addr);
  unsigned int old = *a, assumed;
  do {
    assumed = old;
    old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) * v));
  } while (old != assumed);
}

// values (R, C, N); index (R, Ci, N) int64, Ci in {1, C}; canvas (R, C, M)
__global__ void __launch_bounds__(256)
k_scatter_flat(const float* __restrict__ values, const long long* __restrict__ index,
               float* __restrict__ canvas, float* __restrict__ count, int C, int Ci, size_t N,
               size_t M, int reduction) {
  const int rc = blockIdx.y;                 // r * C + c
  const int r = rc / C, c = rc - r * C;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const long long cell = index[((size_t)r * Ci + (Ci == 1 ? 0 : c)) * N + i];
  if (cell < 0 || (unsigned long long)cell >= M) return;
  const float v = values[(size_t)rc * N + i];
  float* dst = canvas + (size_t)rc * M + cell;
  switch (reduction) {
    case DM_REDUCE_MAX: atomic_max_f(dst, v); break;
    case DM_REDUCE_MIN: atomic_min_f(dst, v); break;
    case DM_REDUCE_PROD: atomic_mul_f(dst, v); break;
    default: atomicAdd(dst, v); break;
  }
  if (count) atomicAdd(count + (size_t)rc * M + cell, 1.0f);
}

// mask = changed (vs the constant fill or vs the saved copy); mean division
__global__ void __launch_bounds__(256)
k_scatter_finalize(float* __restrict__ canvas, const float* __restrict__ before, float fill,
                   const float* __restrict__ count, uint8_t* __restrict__ mask, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float o = canvas[i];
    if (count) {
      const float c = count[i];
      o = o / (c < 1.0f ? 1.0f : c);
      canvas[i] = o;
    }
    mask[i] = mask_of(o, before ? before[i] : fill);
  }
}

__global__ void __launch_bounds__(256)
k_fill1(float* __restrict__ a, float v, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) a[i] = v;
}

// camera_affine_grid (maps.py:353-460): where does every pixel of frame t land
// in the image of frame t+1 after the camera moved by trans_pose?  Pure map
// kernel: 4 bytes in, 8 bytes out per pixel.  Chain (all float32, reference op
// order): image_to_camera_space (616-682) -> camera_to_local_space (753-800)
// -> local_to_global_space(trans_pose) (850-895) -> local_to_camera_space
// (802-848: translate by (0,-h,0), rotate by -pitch) -> camera_to_image_space
// (684-751: z_eps = z + 1e-7, x/z_eps*fx + cx, flip).
__global__ void __launch_bounds__(256)
k_camera_affine_grid(View v, int dc, const dm_frame* __restrict__ frames,
                     const float* __restrict__ depth, float* __restrict__ grid) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int N = v.H * v.W;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const dm_frame* f = frames + b;
  const float* rp = f->Rp; const float* ry = f->Ry; const float* ri = f->reserved;  // R(-pitch)
  const int r = i / v.W, q = i - r * v.W;
  const size_t k = ((size_t)b * dc + ch) * N + i;
  const float z = depth[k];
  const float X = ray_x(v, q) * z, Y = ray_y(v, r) * z;
  float rp9[9], ry9[9], ri9[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { rp9[j] = rp[j]; ry9[j] = ry[j]; ri9[j] = ri[j]; }
  float u, w;
  flow_pixel(z, X, Y, rp9, f->cam_height, ry9, f->tx, f->tz, ri9, v.fx, v.cx, v.fy, v.cy, v.flip_h != 0, v.Hm1, u, w);
  reinterpret_cast<float2*>(grid)[k] = make_float2(u, w);
}

// What the ego-motion kernel reads of a frame, passed in the kernel arguments (no staged copy
// for batches of up to kGridFrames frames).
struct GridFrame { float rp[9], cam_h, ry[9], tx, tz, ri[9], pad; };
constexpr int kGridFrames = 24;          // 24 * 128 bytes of kernel arguments
struct GridFrames { GridFrame f[kGridFrames]; };

// The same, four pixels of a row per trip (W % 4 == 0, 16-byte aligned images): one 16-byte depth
// load and two 16-byte grid stores per trip, two trips in flight, a fixed number of workgroups per
// image that stride over it (a workgroup per 1024 pixels was twenty thousand short-lived blocks
// for a 16-frame batch).  (x - cx) / fx by the exact reciprocal-FMA division of dm_pixel.hpp where
// fx allows it; the two perspective divisions are IEEE divisions.
template <bool FAST_DIV, bool FROM_ARGS>
__global__ void __launch_bounds__(256)
k_camera_affine_grid4(View v, float fx_inv, float fy_inv, int dc, GridFrames args,
                      const dm_frame* __restrict__ frames, int b0, const float* __restrict__ depth,
                      float* __restrict__ grid) {
  const int bl = blockIdx.z, b = b0 + bl, ch = blockIdx.y;
  const int W4 = v.W >> 2;
  const int total = v.H * W4;
  // the frame's record: wave-uniform scalars either way
  float rp[9], ry[9], ri[9], cam_h, tx, tz;
  if (FROM_ARGS) {
    const GridFrame& f = args.f[bl];
#pragma unroll
    for (int i = 0; i < 9; ++i) { rp[i] = f.rp[i]; ry[i] = f.ry[i]; ri[i] = f.ri[i]; }
    cam_h = f.cam_h; tx = f.tx; tz = f.tz;
  } else {
    const dm_frame* f = frames + b;
#pragma unroll
    for (int i = 0; i < 9; ++i) { rp[i] = f->Rp[i]; ry[i] = f->Ry[i]; ri[i] = f->reserved[i]; }
    cam_h = f->cam_height; tx = f->tx; tz = f->tz;
  }
  const size_t base = ((size_t)b * dc + ch) * ((size_t)v.H * v.W);
  // the coefficients as {c, c} pairs (flow_pixel2).  Two pixels per instruction: 85 -> 58 VALU instructions
  // per pixel, 61 -> 46 us at 16 x 1280x960 with the working set beyond the Infinity Cache.  (Pinned in
  // vector registers the pairs need no scalar moves but 128 registers per thread instead of 60: half the
  // waves per SIMD, 69-71 us.)
  FlowPairs c;
  auto pair = [](float x) { return (f32x2){x, x}; };
#pragma unroll
  for (int i = 0; i < 9; ++i) { c.rp[i] = pair(rp[i]); c.ry[i] = pair(ry[i]); c.ri[i] = pair(ri[i]); }
  c.cam_h = pair(cam_h); c.neg_cam_h = pair(-cam_h); c.tx = pair(tx); c.tz = pair(tz);
  c.fx = pair(v.fx); c.cx = pair(v.cx); c.fy = pair(v.fy); c.cy = pair(v.cy);
  c.zero = pair(0.0f); c.eps = pair(1e-7f); c.Hm1 = pair(v.Hm1);
  // chunk i of the image = columns 4 q4 ... 4 q4 + 3 of row r; (r, q4) walk with i (no division per trip)
  auto project4 = [&](int r, int q4, const float4 zz) {
    const int q0 = q4 << 2;
    const float zs[4] = {zz.x, zz.y, zz.z, zz.w};
    float yr = (float)r;
    if (v.flip_h) yr = v.Hm1 - yr;
    const float ay = div_f<FAST_DIV>(yr - v.cy, v.fy, fy_inv);          // maps.py:670-678
    float out[8];
#pragma unroll
    for (int j = 0; j < 4; j += 2) {       // two pixels per instruction
      const f32x2 z = {zs[j], zs[j + 1]};
      const f32x2 ax = {div_f<FAST_DIV>((float)(q0 + j) - v.cx, v.fx, fx_inv),
                        div_f<FAST_DIV>((float)(q0 + j + 1) - v.cx, v.fx, fx_inv)};
      const f32x2 X = ax * z, Y = (f32x2){ay, ay} * z;
      f32x2 u, w;
      flow_pixel2(z, X, Y, c, v.flip_h != 0, u, w);
      out[2 * j] = u.x; out[2 * j + 1] = w.x; out[2 * j + 2] = u.y; out[2 * j + 3] = w.y;
    }
    float4* dst = reinterpret_cast<float4*>(grid + 2 * (base + (size_t)r * v.W + q0));
    // (write-once output, read by nobody in this kernel: non-temporal stores, 48-53 -> 46.6-47.1 us at
    // 16 x 1280x960; non-temporal LOADS of the depth maps on top: 66 us)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((f32x4){out[0], out[1], out[2], out[3]}, reinterpret_cast<f32x4*>(dst));
    __builtin_nontemporal_store((f32x4){out[4], out[5], out[6], out[7]}, reinterpret_cast<f32x4*>(dst) + 1);
  };
  // a workgroup owns one contiguous piece of the image (chunked streams write faster on this chip
  // than grid-strided ones: profiles/r01_microbench.log), its threads stride over it, two trips in flight
  const float4* src = reinterpret_cast<const float4*>(depth + base);
  const int per_block = (total + (int)gridDim.x - 1) / (int)gridDim.x;
  const int end = min(total, ((int)blockIdx.x + 1) * per_block);
  int i = (int)blockIdx.x * per_block + (int)threadIdx.x;
  constexpr int kStep = 256;
  const int step_r = kStep / W4, step_q = kStep - step_r * W4;          // (scalar) one step of kStep chunks in rows / columns
  int r = i / W4, q4 = i - r * W4;
  auto advance = [&](int& rr, int& qq) {
    qq += step_q; rr += step_r;
    const bool wrap = qq >= W4;
    qq -= wrap ? W4 : 0; rr += wrap ? 1 : 0;
  };
  for (; i + kStep < end; i += 2 * kStep) {
    const float4 za = src[i], zb = src[i + kStep];
    int r1 = r, q1 = q4;
    advance(r1, q1);
    project4(r, q4, za);
    project4(r1, q1, zb);
    r = r1; q4 = q1;
    advance(r, q4);
  }
  if (i < end) project4(r, q4, src[i]);
}

inline int nblocks(size_t n, int cap = 8192) {
  size_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > (size_t)cap ? cap : b));
}

}  // namespace

hipError_t run_affine_points(const float* pts, const float* R, const float* t, int B,
                             size_t n_per_batch, int translate_first, float* out, hipStream_t s) {
  if (B == 0 || n_per_batch == 0) return hipSuccess;
  dim3 g((unsigned)((n_per_batch + 255) / 256), B);
  hipLaunchKernelGGL(k_affine_points, g, dim3(256), 0, s, pts, R, t, translate_first,
                     n_per_batch, out);
  return hipGetLastError();
}

hipError_t run_map_quantize(const float* x, const float* z, const float* woff, const float* hoff,
                            int B, size_t n_per_batch, float res, int map_height, int flip,
                            long long* xb, long long* zb, hipStream_t s) {
  if (B == 0 || n_per_batch == 0) return hipSuccess;
  dim3 g((unsigned)((n_per_batch + 255) / 256), B);
  hipLaunchKernelGGL(k_map_quantize, g, dim3(256), 0, s, x, z, woff, hoff, res,
                     (float)(map_height - 1), flip, n_per_batch, xb, zb);
  return hipGetLastError();
}

hipError_t run_camera_affine_grid(const dm_params& p, const dm_frame* frames_host,
                                  const float* depth, float* grid, void* ws, hipStream_t s) {
  if (p.B == 0) return hipSuccess;
  const View v = make_view(p);
  if (p.W % 4 == 0 && reinterpret_cast<uintptr_t>(depth) % 16 == 0 &&
      reinterpret_cast<uintptr_t>(grid) % 16 == 0) {
    float fx_inv = 0.0f, fy_inv = 0.0f;
    const bool fast = exact_reciprocal(p.fx, &fx_inv) && exact_reciprocal(p.fy, &fy_inv) && p.fx >= 1e-6f &&
                      p.fx <= 1e6f && p.fy >= 1e-6f && p.fy <= 1e6f;
    // about sixteen workgroups per CU over the whole launch (measured at 16 x 1280x960: 512 ... 16 K
    // workgroups 61, 57, 52.5, 48, 53, 69 us), whole frames per launch
    const int chunks = p.H * (p.W / 4);
    thread_local GridFrames args;
    for (int b0 = 0; b0 < p.B; b0 += kGridFrames) {
      const int nb = p.B - b0 < kGridFrames ? p.B - b0 : kGridFrames;
      constexpr int target = 4096;
      int per_image = target / (nb * p.dc);
      const int most = (chunks + 255) / 256;
      if (per_image < 1) per_image = 1;
      if (per_image > most) per_image = most;
      for (int i = 0; i < nb; ++i) {
        const dm_frame& f = frames_host[b0 + i];
        GridFrame& g = args.f[i];
        memcpy(g.rp, f.Rp, sizeof(g.rp)); memcpy(g.ry, f.Ry, sizeof(g.ry)); memcpy(g.ri, f.reserved, sizeof(g.ri));
        g.cam_h = f.cam_height; g.tx = f.tx; g.tz = f.tz; g.pad = 0.0f;
      }
      const dim3 g((unsigned)per_image, p.dc, nb);
      if (fast)
        hipLaunchKernelGGL((k_camera_affine_grid4<true, true>), g, dim3(256), 0, s, v, fx_inv, fy_inv, p.dc, args,
                           (const dm_frame*)nullptr, b0, depth, grid);
      else
        hipLaunchKernelGGL((k_camera_affine_grid4<false, true>), g, dim3(256), 0, s, v, fx_inv, fy_inv, p.dc, args,
                           (const dm_frame*)nullptr, b0, depth, grid);
    }
  } else {
    hipError_t e = hipMemcpyAsync(ws, frames_host, (size_t)p.B * sizeof(dm_frame),
                                  hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    dim3 g((unsigned)((p.H * p.W + 255) / 256), p.dc, p.B);
    hipLaunchKernelGGL(k_camera_affine_grid, g, dim3(256), 0, s, v, p.dc,
                       static_cast<const dm_frame*>(ws), depth, grid);
  }
  return hipGetLastError();
}

size_t scatter_workspace_bytes(size_t rows, size_t M, int has_fill, int reduction) {
  size_t n = 0;
  if (!has_fill) n += rows * M * sizeof(float);
  if (reduction == DM_REDUCE_MEAN) n += rows * M * sizeof(float);
  return n;
}

hipError_t run_scatter(const float* values, const long long* index, float* canvas, uint8_t* mask,
                       int R, int C, int Ci, size_t N, size_t M, float fill, int has_fill,
                       int reduction, void* ws, hipStream_t s) {
  const size_t rows = (size_t)R * C, n = rows * M;
  if (n == 0) return hipSuccess;
  float* before = nullptr;
  float* count = nullptr;
  float* w = static_cast<float*>(ws);
  if (has_fill) {
    hipLaunchKernelGGL(k_fill1, dim3(nblocks(n)), dim3(256), 0, s, canvas, fill, n);
  } else {
    before = w; w += n;
    hipError_t e = hipMemcpyAsync(before, canvas, n * sizeof(float), hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
  }
  if (reduction == DM_REDUCE_MEAN) {
    count = w;
    hipLaunchKernelGGL(k_fill1, dim3(nblocks(n)), dim3(256), 0, s, count, 0.0f, n);
  }
  if (N > 0) {
    dim3 g((unsigned)((N + 255) / 256), (unsigned)rows);
    hipLaunchKernelGGL(k_scatter_flat, g, dim3(256), 0, s, values, index, canvas, count, C, Ci, N,
                       M, reduction);
  }
  hipLaunchKernelGGL(k_scatter_finalize, dim3(nblocks(n)), dim3(256), 0, s, canvas, before, fill,
                     count, mask, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// crop_topdown_map / TopdownMap.select (maps.py:1959-2037): generate_crop_grid
// (utils.py:571-611) + image_sample(mode='nearest') (utils.py:613-652) as ONE gather.
//
// The reference pads the image by one pixel per side with the fill value, builds a
// normalised grid and calls grid_sample(nearest, align_corners=True).  Per output pixel
// (i, j) of frame b, float32, one rounding per operation, in the reference's order:
//   c   = center[b] + 1                         pw = w + 2, ph = h + 2
//   gx  = ((j - cw/2) + (c.x - pw/2)) / (pw/2)                         utils.py:603-609
//   ix  = ((gx + 1) / 2) * (pw - 1)             grid_sample, align_corners=True
//   has_fill: ix = clamp(ix, 0, pw - 1)         padding_mode='border'  utils.py:639
//   ixn = nearbyint(ix)                         round half to even
//   out = inside the padded image ? (inside the image ? src : fill) : 0   ('zeros')
// and the same for y.  `mask` (bool image, fill False) rides on the same coordinates.
__global__ void __launch_bounds__(256)
k_crop_nearest(const float* __restrict__ src, const uint8_t* __restrict__ src_mask,
               const float* __restrict__ center, int C, int h, int w, int ch, int cw, float fill,
               int has_fill, float* __restrict__ dst, uint8_t* __restrict__ dst_mask) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ch * cw) return;
  const int i = o / cw, j = o - i * cw;
  const float pw = (float)(w + 2), ph = (float)(h + 2);
  const float cxp = center[2 * b] + 1.0f, cyp = center[2 * b + 1] + 1.0f;
  const float gx = (((float)j - (float)cw / 2.0f) + (cxp - pw / 2.0f)) / (pw / 2.0f);
  const float gy = (((float)i - (float)ch / 2.0f) + (cyp - ph / 2.0f)) / (ph / 2.0f);
  float ix = ((gx + 1.0f) / 2.0f) * (pw - 1.0f);
  float iy = ((gy + 1.0f) / 2.0f) * (ph - 1.0f);
  if (has_fill) {            // clip_coordinates: min(size - 1, max(coord, 0)); NaN -> 0
    ix = fminf(pw - 1.0f, fmaxf(ix, 0.0f));
    iy = fminf(ph - 1.0f, fmaxf(iy, 0.0f));
  }
  const float fx = nearbyintf(ix), fy = nearbyintf(iy);
  const size_t plane = ((size_t)b * C + c);
  float v = has_fill ? fill : 0.0f;
  uint8_t m = 0;
  // inside the original image (padded coordinates 1 .. w / 1 .. h)?
  if (fx >= 1.0f && fx <= (float)w && fy >= 1.0f && fy <= (float)h) {
    const size_t at = plane * h * w + (size_t)((int)fy - 1) * w + ((int)fx - 1);
    v = src[at];
    if (src_mask) m = src_mask[at];
  } else if (!(fx >= 0.0f && fx <= pw - 1.0f && fy >= 0.0f && fy <= ph - 1.0f)) {
    v = 0.0f;                // outside the padded image ('zeros' mode only; NaN coordinates)
  }
  const size_t to = plane * ch * cw + o;
  dst[to] = v;
  if (dst_mask) dst_mask[to] = m;
}

// The same gather, four cells of a crop row per thread (cw % 4 == 0, 16- / 4-byte aligned outputs):
// the row coordinate once per thread, one 16-byte store of the values and one 4-byte store of the
// mask bytes instead of four of each -- the scalar kernel's one-byte mask stores set its pace
// (0.35 of the HBM roofline for a 1024 x 1024 crop of sixteen 2048 x 2048 maps; this form: see
// DESIGN).  Per cell the arithmetic above, operation for operation.
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };     // 16 bytes at 4-byte alignment
struct __attribute__((packed, aligned(1))) U4u { uint32_t v; };            // 4 bytes at any alignment
// ROWS crop rows per thread (rows i, i + rows_per, ...: the same four columns, their coordinates
// computed once), every load of the thread issued before the first is used.
template <int ROWS>
__global__ void __launch_bounds__(256)
k_crop_nearest4(const float* __restrict__ src, const uint8_t* __restrict__ src_mask,
                const float* __restrict__ center, int C, int h, int w, int ch, int cw, float fill,
                int has_fill, float* __restrict__ dst, uint8_t* __restrict__ dst_mask) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int cw4 = cw >> 2;
  const int rows_per = (ch + ROWS - 1) / ROWS;
  const int o4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (o4 >= rows_per * cw4) return;
  const int i0 = o4 / cw4, j0 = (o4 - i0 * cw4) << 2;
  const float pw = (float)(w + 2), ph = (float)(h + 2);
  const float cxp = center[2 * b] + 1.0f, cyp = center[2 * b + 1] + 1.0f;
  const size_t plane = ((size_t)b * C + c);
  // the four source columns first, then every load unconditionally -- a cell outside the map reads
  // column 0 of the row and discards it -- so that all of a thread's loads are in flight together
  // (behind per-cell branches they went out one by one: the kernel waited 73 % of its time)
  int col[4];
  bool in_x[4], pad_x[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gx = (((float)(j0 + k) - (float)cw / 2.0f) + (cxp - pw / 2.0f)) / (pw / 2.0f);
    float ix = ((gx + 1.0f) / 2.0f) * (pw - 1.0f);
    if (has_fill) ix = fminf(pw - 1.0f, fmaxf(ix, 0.0f));
    const float fx = nearbyintf(ix);
    in_x[k] = (fx >= 1.0f) & (fx <= (float)w);
    pad_x[k] = (fx >= 0.0f) & (fx <= pw - 1.0f);
    col[k] = in_x[k] ? (int)fx - 1 : 0;
  }
  const bool run4 = in_x[0] & in_x[3] & (col[1] == col[0] + 1) & (col[2] == col[0] + 2) & (col[3] == col[0] + 3);
  bool row_in[ROWS], row_pad[ROWS], live[ROWS];
  float v[ROWS][4];
  uint32_t mk[ROWS][4];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int i = i0 + r * rows_per;
    live[r] = i < ch;
    const float gy = (((float)i - (float)ch / 2.0f) + (cyp - ph / 2.0f)) / (ph / 2.0f);
    float iy = ((gy + 1.0f) / 2.0f) * (ph - 1.0f);
    if (has_fill) iy = fminf(ph - 1.0f, fmaxf(iy, 0.0f));
    const float fy = nearbyintf(iy);
    row_in[r] = fy >= 1.0f && fy <= (float)h;
    row_pad[r] = fy >= 0.0f && fy <= ph - 1.0f;
    const size_t row_at = plane * h * w + (size_t)(row_in[r] ? (int)fy - 1 : 0) * w;
    if (run4) {
      // four consecutive source columns (all but the few groups where the sampling grid's scale
      // (w + 1) / (w + 2) repeats a column, and those at the map's edge): ONE 16-byte load of the values
      // and one 4-byte load of the mask bytes, at whatever alignment the crop's offset gives them
      const F4u t = *reinterpret_cast<const F4u*>(src + row_at + col[0]);
      v[r][0] = t.x; v[r][1] = t.y; v[r][2] = t.z; v[r][3] = t.w;
      const uint32_t m = src_mask ? reinterpret_cast<const U4u*>(src_mask + row_at + col[0])->v : 0u;
      mk[r][0] = m & 0xffu; mk[r][1] = (m >> 8) & 0xffu; mk[r][2] = (m >> 16) & 0xffu; mk[r][3] = m >> 24;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[r][k] = src[row_at + col[k]];
        mk[r][k] = src_mask ? src_mask[row_at + col[k]] : 0u;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if (!live[r]) continue;
    uint32_t m4 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool in = row_in[r] & in_x[k], pad = row_pad[r] & pad_x[k];
      v[r][k] = in ? v[r][k] : (pad ? (has_fill ? fill : 0.0f) : 0.0f);
      m4 |= (in ? mk[r][k] : 0u) << (8 * k);
    }
    const size_t to = plane * ch * cw + (size_t)(i0 + r * rows_per) * cw + j0;
    *reinterpret_cast<float4*>(dst + to) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
    if (dst_mask) *reinterpret_cast<uint32_t*>(dst_mask + to) = m4;
  }
}

// crop_topdown_map with the interpolating modes: generate_crop_grid + image_sample(mode='bilinear' | 'bicubic')
// (utils.py:571-652) as torch's grid_sample evaluates them on the image padded by one pixel per side
// (align_corners=True; padding 'border' with the pad = fill, or 'zeros' for fill None), one thread per cell:
//   bilinear: coordinates clipped for 'border'; the four neighbours weighted (x_e - ix)(y_s - iy), ..., summed
//             nw, ne, sw, se; neighbours outside the padded image contribute zero;
//   bicubic:  un-clipped coordinates, t = ix - floor(ix), cubic-convolution coefficients (A = -0.75), every tap's
//             coordinate clipped ('border') or bounds-checked ('zeros') on its own; rows first, then the column.
// Float32, one rounding per operation (no contraction): equal to torch's CPU kernel within a few ulp (its vector
// path sums in another order), the same NaN / inf where an empty (-inf) cell meets a zero weight.  A bool image
// (the companion mask, fill False) is sampled as 0 / 1 and written back as "nonzero".
template <int MODE, bool IS_MASK>      // MODE 1: bilinear, 2: bicubic
__global__ void __launch_bounds__(256)
k_crop_interp(const void* __restrict__ src_, const float* __restrict__ center, int C, int h, int w, int ch, int cw,
              float fill, int has_fill, void* __restrict__ dst_) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= ch * cw) return;
  const int i = o / cw, j = o - i * cw;
  const float pw = (float)(w + 2), ph = (float)(h + 2);
  const float cxp = center[2 * b] + 1.0f, cyp = center[2 * b + 1] + 1.0f;
  const float gx = (((float)j - (float)cw / 2.0f) + (cxp - pw / 2.0f)) / (pw / 2.0f);
  const float gy = (((float)i - (float)ch / 2.0f) + (cyp - ph / 2.0f)) / (ph / 2.0f);
  float ix = ((gx + 1.0f) / 2.0f) * (pw - 1.0f);
  float iy = ((gy + 1.0f) / 2.0f) * (ph - 1.0f);
  const size_t plane = ((size_t)b * C + c) * (size_t)h * w;
  const float pad = has_fill ? fill : 0.0f;
  // value of the padded image at integral float coordinates; outside the padded image: 0
  auto tap = [&](float px, float py) -> float {
    if (!(px >= 0.0f && px <= pw - 1.0f && py >= 0.0f && py <= ph - 1.0f)) return 0.0f;
    if (px >= 1.0f && px <= (float)w && py >= 1.0f && py <= (float)h) {
      const size_t at = plane + (size_t)((int)py - 1) * w + ((int)px - 1);
      return IS_MASK ? (static_cast<const uint8_t*>(src_)[at] ? 1.0f : 0.0f) : static_cast<const float*>(src_)[at];
    }
    return pad;
  };
  auto clip = [&](float v, float hi) { return fminf(hi, fmaxf(v, 0.0f)); };
  float out;
  if (MODE == 1) {
    if (has_fill) { ix = clip(ix, pw - 1.0f); iy = clip(iy, ph - 1.0f); }
    const float x_w = floorf(ix), y_n = floorf(iy), x_e = x_w + 1.0f, y_s = y_n + 1.0f;
    const float nw = (x_e - ix) * (y_s - iy), ne = (ix - x_w) * (y_s - iy);
    const float sw = (x_e - ix) * (iy - y_n), se = (ix - x_w) * (iy - y_n);
    out = tap(x_w, y_n) * nw;
    out = out + tap(x_e, y_n) * ne;
    out = out + tap(x_w, y_s) * sw;
    out = out + tap(x_e, y_s) * se;
  } else {
    constexpr float A = -0.75f;
    auto conv1 = [&](float x) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; };
    auto conv2 = [&](float x) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; };
    const float x_nw = floorf(ix), y_nw = floorf(iy);
    const float tx = ix - x_nw, ty = iy - y_nw, ux = 1.0f - tx, uy = 1.0f - ty;
    const float kx[4] = {conv2(tx + 1.0f), conv1(tx), conv1(ux), conv2(ux + 1.0f)};
    const float ky[4] = {conv2(ty + 1.0f), conv1(ty), conv1(uy), conv2(uy + 1.0f)};
    out = 0.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float py = y_nw - 1.0f + (float)r;
      if (has_fill) py = clip(py, ph - 1.0f);
      float row = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float px = x_nw - 1.0f + (float)q;
        if (has_fill) px = clip(px, pw - 1.0f);
        const float term = tap(px, py) * kx[q];
        row = q == 0 ? term : row + term;
      }
      const float term = row * ky[r];
      out = r == 0 ? term : out + term;
    }
  }
  const size_t to = ((size_t)b * C + c) * (size_t)ch * cw + o;
  if (IS_MASK) static_cast<uint8_t*>(dst_)[to] = out != 0.0f ? 1 : 0;
  else static_cast<float*>(dst_)[to] = out;
}

hipError_t run_crop_interp(const float* src, const uint8_t* src_mask, const float* center, int B, int C, int h, int w,
                           int ch, int cw, float fill, int has_fill, int mode, float* dst, uint8_t* dst_mask,
                           hipStream_t s) {
  const dim3 grid((unsigned)(((size_t)ch * cw + 255) / 256), C, B);
  if (mode == 1) {
    hipLaunchKernelGGL((k_crop_interp<1, false>), grid, dim3(256), 0, s, src, center, C, h, w, ch, cw, fill, has_fill, dst);
    if (src_mask)      // (the mask: fill False, 'border')
      hipLaunchKernelGGL((k_crop_interp<1, true>), grid, dim3(256), 0, s, src_mask, center, C, h, w, ch, cw, 0.0f, 1, dst_mask);
  } else {
    hipLaunchKernelGGL((k_crop_interp<2, false>), grid, dim3(256), 0, s, src, center, C, h, w, ch, cw, fill, has_fill, dst);
    if (src_mask)
      hipLaunchKernelGGL((k_crop_interp<2, true>), grid, dim3(256), 0, s, src_mask, center, C, h, w, ch, cw, 0.0f, 1, dst_mask);
  }
  return hipGetLastError();
}

hipError_t run_crop_nearest(const float* src, const uint8_t* src_mask, const float* center, int B,
                            int C, int h, int w, int ch, int cw, float fill, int has_fill,
                            float* dst, uint8_t* dst_mask, hipStream_t s) {
  if (cw % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0 && reinterpret_cast<uintptr_t>(dst_mask) % 4 == 0) {
    constexpr int kRows = 1;       // (two / four crop rows per thread: no better / worse)
    const dim3 grid((unsigned)(((size_t)((ch + kRows - 1) / kRows) * (cw / 4) + 255) / 256), C, B);
    hipLaunchKernelGGL(k_crop_nearest4<kRows>, grid, dim3(256), 0, s, src, src_mask, center, C, h, w, ch, cw,
                       fill, has_fill, dst, dst_mask);
    return hipGetLastError();
  }
  const dim3 grid((unsigned)(((size_t)ch * cw + 255) / 256), C, B);
  hipLaunchKernelGGL(k_crop_nearest, grid, dim3(256), 0, s, src, src_mask, center, C, h, w, ch, cw,
                     fill, has_fill, dst, dst_mask);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// fuse_topdown_maps (maps.py:2039-2287): see include/dungeon_maps_amd.h
// ---------------------------------------------------------------------------
namespace {

// cell (row, col) of batch row bi -> its point in the target's frame
__device__ inline void fuse_point(const dm_fuse_src& s, int bi, int row, int col, float y,
                                  float& xo, float& yo, float& zo) {
  float r = (float)row;
  if (s.flip_h) r = (float)(s.h - 1) - r;                       // maps.py:1074-1076
  float p0 = ((float)col - s.woff[bi]) * s.res;                  // maps.py:1078-1079
  float p1 = y;
  float p2 = (r - s.hoff[bi]) * s.res;
  if (s.has_l2g) {                                               // rotate, then translate
    const float* m = s.l2g[bi];
    const float o0 = __builtin_fmaf(p2, m[6], __builtin_fmaf(p1, m[3], p0 * m[0])) + m[9];
    const float o1 = __builtin_fmaf(p2, m[7], __builtin_fmaf(p1, m[4], p0 * m[1])) + m[10];
    const float o2 = __builtin_fmaf(p2, m[8], __builtin_fmaf(p1, m[5], p0 * m[2])) + m[11];
    p0 = o0; p1 = o1; p2 = o2;
  }
  if (s.has_g2l) {                                               // translate, then rotate
    const float* m = s.g2l[bi];
    p0 += m[9]; p1 += m[10]; p2 += m[11];
    const float o0 = __builtin_fmaf(p2, m[6], __builtin_fmaf(p1, m[3], p0 * m[0]));
    const float o1 = __builtin_fmaf(p2, m[7], __builtin_fmaf(p1, m[4], p0 * m[1]));
    const float o2 = __builtin_fmaf(p2, m[8], __builtin_fmaf(p1, m[5], p0 * m[2]));
    p0 = o0; p1 = o1; p2 = o2;
  }
  xo = p0; yo = p1; zo = p2;
}

__device__ inline int sat_i32(float f) {      // floor result -> int32, saturating; NaN -> INT_MIN
  if (!(f == f)) return (int)0x80000000;
  if (f >= 2147483520.0f) return 0x7fffffff;
  if (f <= -2147483648.0f) return (int)0x80000000;
  return (int)f;
}

__global__ void k_fuse_stats_init(int* stats) {
  if (threadIdx.x == 0) {
    stats[0] = 0x7fffffff; stats[1] = (int)0x80000000;
    stats[2] = 0x7fffffff; stats[3] = (int)0x80000000; stats[4] = 0;
  }
}

__global__ void __launch_bounds__(256)
k_fuse_bbox(dm_fuse_src s, int* __restrict__ stats) {
  __shared__ int sh[5];
  if (threadIdx.x < 5)
    sh[threadIdx.x] = (threadIdx.x == 0 || threadIdx.x == 2) ? 0x7fffffff
                    : (threadIdx.x == 4 ? 0 : (int)0x80000000);
  __syncthreads();
  const int bi = blockIdx.z, ci = blockIdx.y;
  const int n = s.h * s.w;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const size_t plane = (size_t)s.h * s.w;
    if (s.mask_dev[((size_t)bi * s.mc + (s.mc == 1 ? 0 : ci)) * plane + i]) {
      const int row = i / s.w, col = i - row * s.w;
      const float y = s.height_dev[((size_t)bi * s.hc + (s.hc == 1 ? 0 : ci)) * plane + i];
      float x, yy, z;
      fuse_point(s, bi, row, col, y, x, yy, z);
      // map_quantize with zero offsets, unflipped (maps.py:2146-2160)
      const int c0 = sat_i32(__builtin_floorf((x / s.target_res + 0.0f) + 0.5f));
      const int r0 = sat_i32(__builtin_floorf((z / s.target_res + 0.0f) + 0.5f));
      atomicMin(&sh[0], c0); atomicMax(&sh[1], c0);
      atomicMin(&sh[2], r0); atomicMax(&sh[3], r0);
      sh[4] = 1;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && sh[4]) {
    atomicMin(&stats[0], sh[0]); atomicMax(&stats[1], sh[1]);
    atomicMin(&stats[2], sh[2]); atomicMax(&stats[3], sh[3]);
    atomicOr(&stats[4], 1);
  }
}

__global__ void __launch_bounds__(256)
k_fuse_scatter(dm_fuse_src s, float woff, float hoff, int flip, float mhm1, int mh, int mw,
               int is_max, float* __restrict__ canvas, float* __restrict__ hcanvas) {
  const int bi = blockIdx.z, ci = blockIdx.y;
  const int n = s.h * s.w;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t plane = (size_t)s.h * s.w;
  if (!s.mask_dev[((size_t)bi * s.mc + (s.mc == 1 ? 0 : ci)) * plane + i]) return;
  const int row = i / s.w, col = i - row * s.w;
  const float y = s.height_dev[((size_t)bi * s.hc + (s.hc == 1 ? 0 : ci)) * plane + i];
  float x, yy, z;
  fuse_point(s, bi, row, col, y, x, yy, z);
  const float xf = x / s.target_res + woff;                     // maps.py:1004-1015
  float zf = z / s.target_res + hoff;
  if (flip) zf = mhm1 - zf;
  const float cf = __builtin_floorf(xf + 0.5f), rf = __builtin_floorf(zf + 0.5f);
  if (!(cf >= 0.0f && cf < (float)mw && rf >= 0.0f && rf < (float)mh)) return;   // maps.py:1150-1158
  const size_t cell = (size_t)(int)rf * mw + (int)cf;
  const size_t M = (size_t)mh * mw;
  const float v = s.value_dev ? s.value_dev[((size_t)bi * s.c + ci) * plane + i] : yy;
  float* dst = canvas + ((size_t)bi * s.c + ci) * M + cell;
  if (is_max) atomic_max_f(dst, v); else atomic_min_f(dst, v);
  // (per channel: a cell counts where THIS channel's mask is set, maps.py:2257-2272)
  if (hcanvas) atomic_max_f(hcanvas + ((size_t)bi * s.c + ci) * M + cell, yy);
}

constexpr int kBboxThreads = 1024;            // k_fuse_bbox_multi
constexpr int kBboxCellsPerThread = 8;        // ... cells per thread and pass (blocks per plane capped at kBboxMaxBlocks)
constexpr size_t kBboxMaxBlocks = 64;

// Several source maps in one launch (MapBuilder.merge fuses two maps per frame: the launches
// were a third of its device time).  blockIdx.z = source * b + batch row.
struct FuseSources {
  int n;
  dm_fuse_src s[DM_FUSE_MAX_SOURCES];
};

// The bounding box as FIVE MAXIMA of unsigned words that start at zero (so a zero-filled stats
// buffer needs no initialising launch): u = x ^ 0x80000000 orders like x;
// stats = {max ~u(col), max u(col), max ~u(row), max u(row), any}.
__global__ void __launch_bounds__(kBboxThreads)
k_fuse_bbox_multi(FuseSources a, unsigned* __restrict__ stats) {
  __shared__ unsigned sh[5];
  if (threadIdx.x < 5) sh[threadIdx.x] = 0u;
  __syncthreads();
  const int si = blockIdx.z / a.s[0].b, bi = blockIdx.z - si * a.s[0].b, ci = blockIdx.y;
  const dm_fuse_src& s = a.s[si];
  const int n = s.h * s.w;
  const size_t plane = (size_t)s.h * s.w;
  const uint8_t* mask = s.mask_dev + ((size_t)bi * s.mc + (s.mc == 1 ? 0 : ci)) * plane;
  const float* height = s.height_dev + ((size_t)bi * s.hc + (s.hc == 1 ? 0 : ci)) * plane;
  // A block walks its share of the plane, kBboxCellsPerThread cells per thread and pass (their mask bytes
  // requested together, then their heights: two round trips per pass, not two per cell), and joins the five words
  // ONCE, wave by wave.  One block per 256 cells put 5 x 512 atomics on ONE cache line for two 256 x 256 maps:
  // 15-18 us of every MapBuilder.merge, all of it that line's round trips (round 5).
  unsigned m0 = 0u, m1 = 0u, m2 = 0u, m3 = 0u;
  for (int base = blockIdx.x * kBboxThreads * kBboxCellsPerThread + threadIdx.x; base < n;
       base += gridDim.x * kBboxThreads * kBboxCellsPerThread) {
    bool ok[kBboxCellsPerThread];
    float y[kBboxCellsPerThread];
#pragma unroll
    for (int u = 0; u < kBboxCellsPerThread; ++u) {
      const int i = base + u * kBboxThreads;
      ok[u] = i < n && mask[i] != 0;
    }
#pragma unroll
    for (int u = 0; u < kBboxCellsPerThread; ++u) {
      const int i = base + u * kBboxThreads;
      y[u] = ok[u] ? height[i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kBboxCellsPerThread; ++u) {
      if (!ok[u]) continue;
      const int i = base + u * kBboxThreads;
      const int row = i / s.w, col = i - row * s.w;
      float x, yy, z;
      fuse_point(s, bi, row, col, y[u], x, yy, z);
      // map_quantize with zero offsets, unflipped (maps.py:2146-2160)
      const unsigned c0 = (unsigned)sat_i32(__builtin_floorf((x / s.target_res + 0.0f) + 0.5f)) ^ 0x80000000u;
      const unsigned r0 = (unsigned)sat_i32(__builtin_floorf((z / s.target_res + 0.0f) + 0.5f)) ^ 0x80000000u;
      m0 = max(m0, ~c0); m1 = max(m1, c0);
      m2 = max(m2, ~r0); m3 = max(m3, r0);
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    m0 = max(m0, (unsigned)__shfl_xor((int)m0, d)); m1 = max(m1, (unsigned)__shfl_xor((int)m1, d));
    m2 = max(m2, (unsigned)__shfl_xor((int)m2, d)); m3 = max(m3, (unsigned)__shfl_xor((int)m3, d));
  }
  // (a wave that saw a valid cell has m0 | m1 != 0: c0 and ~c0 cannot both be zero)
  if ((threadIdx.x & 63) == 0 && (m0 | m1)) {
    atomicMax(&sh[0], m0); atomicMax(&sh[1], m1);
    atomicMax(&sh[2], m2); atomicMax(&sh[3], m3);
    sh[4] = 1u;
  }
  __syncthreads();
  if (threadIdx.x == 0 && sh[4]) {
    atomicMax(&stats[0], sh[0]); atomicMax(&stats[1], sh[1]);
    atomicMax(&stats[2], sh[2]); atomicMax(&stats[3], sh[3]);
    atomicMax(&stats[4], 1u);
  }
}

__global__ void __launch_bounds__(256)
k_fuse_scatter_multi(FuseSources a, float woff, float hoff, int flip, float mhm1, int mh, int mw,
                     int is_max, float* __restrict__ canvas, float* __restrict__ hcanvas) {
  const int si = blockIdx.z / a.s[0].b, bi = blockIdx.z - si * a.s[0].b, ci = blockIdx.y;
  const dm_fuse_src& s = a.s[si];
  const int n = s.h * s.w;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t plane = (size_t)s.h * s.w;
  if (!s.mask_dev[((size_t)bi * s.mc + (s.mc == 1 ? 0 : ci)) * plane + i]) return;
  const int row = i / s.w, col = i - row * s.w;
  const float y = s.height_dev[((size_t)bi * s.hc + (s.hc == 1 ? 0 : ci)) * plane + i];
  float x, yy, z;
  fuse_point(s, bi, row, col, y, x, yy, z);
  const float xf = x / s.target_res + woff;                     // maps.py:1004-1015
  float zf = z / s.target_res + hoff;
  if (flip) zf = mhm1 - zf;
  const float cf = __builtin_floorf(xf + 0.5f), rf = __builtin_floorf(zf + 0.5f);
  if (!(cf >= 0.0f && cf < (float)mw && rf >= 0.0f && rf < (float)mh)) return;   // maps.py:1150-1158
  const size_t cell = (size_t)(int)rf * mw + (int)cf;
  const size_t M = (size_t)mh * mw;
  const float v = s.value_dev ? s.value_dev[((size_t)bi * s.c + ci) * plane + i] : yy;
  float* dst = canvas + ((size_t)bi * s.c + ci) * M + cell;
  if (is_max) atomic_max_f(dst, v); else atomic_min_f(dst, v);
  if (hcanvas) atomic_max_f(hcanvas + ((size_t)bi * s.c + ci) * M + cell, yy);
}

inline void gather_sources(const dm_fuse_src* srcs, int n, FuseSources& a, size_t& cells) {
  a.n = n;
  cells = 0;
  for (int i = 0; i < n; ++i) {
    a.s[i] = srcs[i];
    const size_t c = (size_t)srcs[i].h * srcs[i].w;
    cells = c > cells ? c : cells;
  }
}

}  // namespace

hipError_t run_fuse_bbox_multi(const dm_fuse_src* srcs, int n, int* stats, hipStream_t st) {
  FuseSources a;
  size_t cells;
  gather_sources(srcs, n, a, cells);
  size_t blocks = (cells + kBboxThreads * kBboxCellsPerThread - 1) / (kBboxThreads * kBboxCellsPerThread);
  if (blocks > kBboxMaxBlocks) blocks = kBboxMaxBlocks;
  if (blocks < 1) blocks = 1;
  const dim3 g((unsigned)blocks, srcs[0].c, srcs[0].b * n);
  hipLaunchKernelGGL(k_fuse_bbox_multi, g, dim3(kBboxThreads), 0, st, a, reinterpret_cast<unsigned*>(stats));
  return hipGetLastError();
}

hipError_t run_fuse_scatter_multi(const dm_fuse_src* srcs, int n, float woff, float hoff, int flip,
                                  int mh, int mw, int is_max, float* canvas, float* hcanvas,
                                  hipStream_t st) {
  FuseSources a;
  size_t cells;
  gather_sources(srcs, n, a, cells);
  const dim3 g((unsigned)((cells + 255) / 256), srcs[0].c, srcs[0].b * n);
  hipLaunchKernelGGL(k_fuse_scatter_multi, g, dim3(256), 0, st, a, woff, hoff, flip, (float)(mh - 1),
                     mh, mw, is_max, canvas, hcanvas);
  return hipGetLastError();
}

hipError_t run_fuse_bbox(const dm_fuse_src& s, int* stats, int init, hipStream_t st) {
  if (init) hipLaunchKernelGGL(k_fuse_stats_init, dim3(1), dim3(64), 0, st, stats);
  const dim3 g((unsigned)(((size_t)s.h * s.w + 255) / 256), s.c, s.b);
  hipLaunchKernelGGL(k_fuse_bbox, g, dim3(256), 0, st, s, stats);
  return hipGetLastError();
}

hipError_t run_fuse_scatter(const dm_fuse_src& s, float woff, float hoff, int flip, int mh, int mw,
                            int is_max, float* canvas, float* hcanvas, hipStream_t st) {
  const dim3 g((unsigned)(((size_t)s.h * s.w + 255) / 256), s.c, s.b);
  hipLaunchKernelGGL(k_fuse_scatter, g, dim3(256), 0, st, s, woff, hoff, flip, (float)(mh - 1), mh,
                     mw, is_max, canvas, hcanvas);
  return hipGetLastError();
}

}  // namespace dm
