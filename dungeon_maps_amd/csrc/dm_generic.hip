// Generic (always-correct) path: fill -> per-pixel global atomics -> finalize.
//
// Handles every reduction, every map size and every channel layout.  It is the
// fallback behind the LDS-windowed fast path (dm_window.hip) and the reference
// implementation the fast path is tested against on the device.  Memory-side
// atomics make it ~10-50x slower than the roofline, see DESIGN.md.
#include <string.h>

#include <vector>

#include "dm_kernels.hpp"
#include "dm_window_geometry.hpp"      // frame_affine / part_window: a frame's reach in the map

namespace dm {

// ---- float atomics on global memory, native integer/float instructions -----
__device__ inline void atomic_max_f32(float* addr, float v) {
  if (!(v == v)) return;                 // torch_scatter stores only if new > old
  v += 0.0f;                             // -0 -> +0 so the sign test is an order test
  if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ inline void atomic_min_f32(float* addr, float v) {
  if (!(v == v)) return;
  v += 0.0f;
  if (v >= 0.0f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ inline void atomic_mul_f32(float* addr, float v) {
  unsigned int* a = reinterpret_cast<unsigned int*>(addr);
  unsigned int old = *a, assumed;
  do {
    assumed = old;
    old = atomicCAS(a, assumed, __float_as_uint(__uint_as_float(assumed) * v));
  } while (old != assumed);
}

template <int RED>
__device__ inline void reduce_into(float* addr, float v) {
  if (RED == DM_REDUCE_MAX) atomic_max_f32(addr, v);
  else if (RED == DM_REDUCE_MIN) atomic_min_f32(addr, v);
  else if (RED == DM_REDUCE_PROD) atomic_mul_f32(addr, v);
  else atomicAdd(addr, v);               // sum, mean
}

// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_fill(float* __restrict__ a, float va, size_t na, float* __restrict__ b, float vb,
       size_t nb, uint32_t* __restrict__ zero = nullptr, size_t nzero = 0) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < na; i += stride) a[i] = va;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += stride) b[i] = vb;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nzero; i += stride) zero[i] = 0u;
}

// One thread per pixel; grid = (ceil(N/256), dc, B).  FUSED: all frames write
// the same (oc, mh, mw) map.
template <int RED, bool FUSED>
__global__ void __launch_bounds__(256)
k_scatter_generic(View v, int dc, int vc, int valid_c, const dm_frame* __restrict__ frames,
                  const float* __restrict__ depth, const float* __restrict__ value,
                  const uint8_t* __restrict__ valid, float* __restrict__ out,
                  float* __restrict__ height, float* __restrict__ count) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int N = v.H * v.W;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const Cam cam = load_cam(frames + b);
  const int r = i / v.W, q = i - r * v.W;
  bool ok = border_ok(v, r, q);
  if (valid) ok = ok && valid[((size_t)b * valid_c + (valid_c == 1 ? 0 : ch)) * N + i] != 0;
  const float z = depth[((size_t)b * dc + ch) * N + i];
  const Hit h = project_pixel(v, cam, z, ray_x(v, q), ray_y(v, r), ok);
  if (h.cell < 0) return;
  const size_t M = (size_t)v.mh * v.mw;
  const int oc = vc ? vc : dc;
  float* ob = out + (FUSED ? 0 : (size_t)b * oc * M);
  if (!vc) {
    reduce_into<RED>(ob + (size_t)ch * M + h.cell, h.y);
    if (RED == DM_REDUCE_MEAN) atomicAdd(count + ((size_t)b * oc + ch) * M + h.cell, 1.0f);
  } else if (dc == 1) {
    for (int k = 0; k < vc; ++k) {
      reduce_into<RED>(ob + (size_t)k * M + h.cell, value[((size_t)b * vc + k) * N + i]);
      if (RED == DM_REDUCE_MEAN) atomicAdd(count + ((size_t)b * oc + k) * M + h.cell, 1.0f);
    }
  } else {
    reduce_into<RED>(ob + (size_t)ch * M + h.cell, value[((size_t)b * vc + ch) * N + i]);
    if (RED == DM_REDUCE_MEAN) atomicAdd(count + ((size_t)b * oc + ch) * M + h.cell, 1.0f);
  }
  if (height) atomic_max_f32(height + ((size_t)b * dc + ch) * M + h.cell, h.y);
}

// mask (+ mean division).  n = elements of out.  Four cells per thread (16-byte loads, one
// 4-byte mask store: single-byte stores run at a third of the rate) when out / count are
// 16-byte and mask 4-byte aligned, which the host checks; the tail is done cell by cell.
__device__ inline float finalize_cell(float o, float c, bool mean) {
  return mean ? o / (c < 1.0f ? 1.0f : c) : o;
}

template <bool VEC>
__global__ void __launch_bounds__(256)
k_finalize(float* __restrict__ out, const float* __restrict__ count, float fill,
           uint8_t* __restrict__ mask, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n4 = VEC ? n / 4 : 0;
  for (size_t g = first; g < n4; g += stride) {
    float4 o = reinterpret_cast<const float4*>(out)[g];
    if (count) {
      const float4 c = reinterpret_cast<const float4*>(count)[g];
      o.x = finalize_cell(o.x, c.x, true); o.y = finalize_cell(o.y, c.y, true);
      o.z = finalize_cell(o.z, c.z, true); o.w = finalize_cell(o.w, c.w, true);
      reinterpret_cast<float4*>(out)[g] = o;
    }
    reinterpret_cast<uint32_t*>(mask)[g] =
        (uint32_t)mask_of(o.x, fill) | ((uint32_t)mask_of(o.y, fill) << 8) |
        ((uint32_t)mask_of(o.z, fill) << 16) | ((uint32_t)mask_of(o.w, fill) << 24);
  }
  for (size_t i = n4 * 4 + first; i < n; i += stride) {
    float o = out[i];
    if (count) {
      o = finalize_cell(o, count[i], true);
      out[i] = o;
    }
    mask[i] = mask_of(o, fill);
  }
}

static hipError_t launch_finalize(float* out, const float* count, float fill, uint8_t* mask, size_t n,
                                  hipStream_t s) {
  const bool vec = reinterpret_cast<uintptr_t>(out) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(count) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(mask) % 4 == 0;
  const size_t threads = vec ? (n + 3) / 4 : n;
  size_t blocks = (threads + 1023) / 1024;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  if (vec) hipLaunchKernelGGL(k_finalize<true>, dim3((unsigned)blocks), dim3(256), 0, s, out, count, fill, mask, n);
  else hipLaunchKernelGGL(k_finalize<false>, dim3((unsigned)blocks), dim3(256), 0, s, out, count, fill, mask, n);
  return hipGetLastError();
}

// The same restricted to each frame's union window (x0, z0, w, h; x0 and w multiples of 4): on
// large maps at fine resolution a frame reaches a small part of its map, the rest keeps the
// fill value and the mask 0 that k_fill wrote.  grid = (groups of the largest window, B * oc).
__global__ void __launch_bounds__(256)
k_finalize_unions(float* __restrict__ out, const float* __restrict__ count, float fill,
                  uint8_t* __restrict__ mask, const int4* __restrict__ unions, int oc, int mh,
                  int mw) {
  const int b = blockIdx.y / oc;
  const int4 U = unions[b];
  const int ug4 = U.z >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ug4 * U.w) return;
  const int row = i / ug4;
  const size_t cell = (size_t)blockIdx.y * mh * mw + (size_t)(U.y + row) * mw + U.x + ((i - row * ug4) << 2);
  float4 o = *reinterpret_cast<const float4*>(out + cell);
  if (count) {
    const float4 c = *reinterpret_cast<const float4*>(count + cell);
    o.x = finalize_cell(o.x, c.x, true); o.y = finalize_cell(o.y, c.y, true);
    o.z = finalize_cell(o.z, c.z, true); o.w = finalize_cell(o.w, c.w, true);
    *reinterpret_cast<float4*>(out + cell) = o;
  }
  *reinterpret_cast<uint32_t*>(mask + cell) =
      (uint32_t)mask_of(o.x, fill) | ((uint32_t)mask_of(o.y, fill) << 8) |
      ((uint32_t)mask_of(o.z, fill) << 16) | ((uint32_t)mask_of(o.w, fill) << 24);
}

static inline int blocks_for(size_t n, int per_block, int cap) {
  size_t b = (n + per_block - 1) / per_block;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

template <bool FUSED>
static hipError_t launch_scatter(const dm_params& p, const View& v, dim3 grid,
                                 const dm_frame* frames, const float* depth,
                                 const float* value, const uint8_t* valid, float* out,
                                 float* height, float* count, hipStream_t s) {
#define DM_CASE(R)                                                                     \
  case R:                                                                              \
    hipLaunchKernelGGL((k_scatter_generic<R, FUSED>), grid, dim3(256), 0, s, v, p.dc,   \
                       p.vc, p.valid_c, frames, depth, value, valid, out, height, count); \
    break;
  switch (p.reduction) {
    DM_CASE(DM_REDUCE_MAX)
    DM_CASE(DM_REDUCE_MIN)
    DM_CASE(DM_REDUCE_SUM)
    DM_CASE(DM_REDUCE_MEAN)
    DM_CASE(DM_REDUCE_PROD)
    default: return hipErrorInvalidValue;
  }
#undef DM_CASE
  return hipGetLastError();
}

static inline size_t frames_bytes(const dm_params& p) {
  return ((size_t)p.B * sizeof(dm_frame) + 255) / 256 * 256;
}

static inline size_t unions_bytes(const dm_params& p) {
  return ((size_t)p.B * sizeof(int4) + 255) / 256 * 256;
}

// workspace: frame table | union windows | per-cell counts (mean only)
size_t generic_workspace_bytes(const dm_params& p) {
  size_t n = frames_bytes(p) + unions_bytes(p);
  if (p.reduction == DM_REDUCE_MEAN) {
    const size_t oc = p.vc ? p.vc : p.dc;
    n += (size_t)p.B * oc * p.mh * p.mw * sizeof(float);
  }
  return n;
}

// pageable -> device: the bytes have left the host buffer when this returns
static hipError_t upload_frames(const dm_params& p, const dm_frame* frames_host, void* ws,
                                hipStream_t s) {
  return hipMemcpyAsync(ws, frames_host, (size_t)p.B * sizeof(dm_frame), hipMemcpyHostToDevice, s);
}

// Union window of every frame (the map cells its frustum can reach, x aligned to 4); returns
// false when restricting the finalize pass to them cannot work or does not pay (unbounded
// frustum, odd map width, windows covering more than half of the maps).
static bool frame_unions(const dm_params& p, const dm_frame* frames_host, std::vector<int4>& unions,
                         int* max_area) {
  if (p.mw % 4 != 0 || !frustum_bounded(p)) return false;
  const PartSlopes whole = part_slopes(p, 0, p.W, 0, p.H);
  unions.resize(p.B);
  size_t covered = 0;
  *max_area = 0;
  for (int b = 0; b < p.B; ++b) {
    const FrameAffine fa = frame_affine(p, frames_host[b]);
    const Window w = part_window(p, fa, whole, true, p.dmin, p.dmax);
    unions[b] = make_int4(w.x0, w.z0, w.w, w.w > 0 ? w.h : 0);
    covered += (size_t)w.w * w.h;
    if (w.w * w.h > *max_area) *max_area = w.w * w.h;
  }
  return covered * 2 <= (size_t)p.B * p.mh * p.mw;
}

hipError_t run_generic(const dm_params& p, const dm_frame* frames_host, const float* depth,
                       const float* value, const uint8_t* valid, float* out,
                       uint8_t* mask, float* height, void* ws, hipStream_t s) {
  const View v = make_view(p);
  const size_t M = (size_t)p.mh * p.mw, oc = p.vc ? p.vc : p.dc;
  const size_t n_out = (size_t)p.B * oc * M;
  const size_t n_h = height ? (size_t)p.B * p.dc * M : 0;
  unsigned char* base = static_cast<unsigned char*>(ws);
  const dm_frame* frames = static_cast<const dm_frame*>(ws);
  const int4* d_unions = reinterpret_cast<const int4*>(base + frames_bytes(p));
  float* count = p.reduction == DM_REDUCE_MEAN
      ? reinterpret_cast<float*>(base + frames_bytes(p) + unions_bytes(p)) : nullptr;
  // frames (+ union windows) in one stream-ordered copy
  thread_local std::vector<int4> unions;
  thread_local std::vector<unsigned char> staging;
  int max_area = 0;
  const bool by_unions = reinterpret_cast<uintptr_t>(out) % 16 == 0 &&
                         reinterpret_cast<uintptr_t>(mask) % 4 == 0 &&
                         reinterpret_cast<uintptr_t>(ws) % 256 == 0 &&
                         frame_unions(p, frames_host, unions, &max_area);
  hipError_t ue;
  if (by_unions) {
    staging.resize(frames_bytes(p) + (size_t)p.B * sizeof(int4));
    memcpy(staging.data(), frames_host, (size_t)p.B * sizeof(dm_frame));
    memcpy(staging.data() + frames_bytes(p), unions.data(), (size_t)p.B * sizeof(int4));
    ue = hipMemcpyAsync(ws, staging.data(), staging.size(), hipMemcpyHostToDevice, s);
  } else {
    ue = upload_frames(p, frames_host, ws, s);
  }
  if (ue != hipSuccess) return ue;
  hipLaunchKernelGGL(k_fill, dim3(blocks_for(n_out > n_h ? n_out : n_h, 1024, 4096)), dim3(256),
                     0, s, out, p.fill, n_out, height, -__builtin_inff(), n_h,
                     by_unions ? reinterpret_cast<uint32_t*>(mask) : nullptr,
                     by_unions ? n_out / 4 : (size_t)0);
  if (count)
    hipLaunchKernelGGL(k_fill, dim3(blocks_for(n_out, 1024, 4096)), dim3(256), 0, s, count, 0.0f,
                       n_out, (float*)nullptr, 0.0f, (size_t)0, (uint32_t*)nullptr, (size_t)0);
  const int N = p.H * p.W;
  dim3 grid((N + 255) / 256, p.dc, p.B);
  hipError_t e = launch_scatter<false>(p, v, grid, frames, depth, value, valid, out, height,
                                       count, s);
  if (e != hipSuccess) return e;
  if (!by_unions) return launch_finalize(out, count, p.fill, mask, n_out, s);
  if (max_area == 0) return hipSuccess;           // no frame reaches its map
  hipLaunchKernelGGL(k_finalize_unions, dim3((unsigned)((max_area / 4 + 255) / 256), (unsigned)(p.B * oc)),
                     dim3(256), 0, s, out, count, p.fill, mask, d_unions, (int)oc, p.mh, p.mw);
  return hipGetLastError();
}

hipError_t run_generic_fused(const dm_params& p, const dm_frame* frames_host, const float* depth,
                             const float* value, const uint8_t* valid, float* out,
                             uint8_t* mask, int accumulate, void* ws, hipStream_t s) {
  hipError_t ue = upload_frames(p, frames_host, ws, s);
  if (ue != hipSuccess) return ue;
  const dm_frame* frames = static_cast<const dm_frame*>(ws);
  const View v = make_view(p);
  const size_t M = (size_t)p.mh * p.mw, oc = p.vc ? p.vc : p.dc;
  const size_t n_out = oc * M;
  if (!accumulate)
    hipLaunchKernelGGL(k_fill, dim3(blocks_for(n_out, 1024, 4096)), dim3(256), 0, s, out, p.fill,
                       n_out, (float*)nullptr, 0.0f, (size_t)0, (uint32_t*)nullptr, (size_t)0);
  const int N = p.H * p.W;
  dim3 grid((N + 255) / 256, p.dc, p.B);
  hipError_t e = launch_scatter<true>(p, v, grid, frames, depth, value, valid, out, nullptr,
                                      nullptr, s);
  if (e != hipSuccess) return e;
  return launch_finalize(out, nullptr, p.fill, mask, n_out, s);
}

// out[i] = max/min over b of maps[b][i]; 4 cells per thread, 16-byte accesses.
template <bool IS_MAX>
__global__ void __launch_bounds__(256)
k_fuse_batch(const float* __restrict__ maps, int B, size_t n, size_t nvec,
             float* __restrict__ out, int accumulate) {
  const size_t n4 = nvec / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 acc = reinterpret_cast<const float4*>(maps)[i];
    if (accumulate) {
      const float4 o = reinterpret_cast<const float4*>(out)[i];
      acc.x = IS_MAX ? fmaxf(acc.x, o.x) : fminf(acc.x, o.x);
      acc.y = IS_MAX ? fmaxf(acc.y, o.y) : fminf(acc.y, o.y);
      acc.z = IS_MAX ? fmaxf(acc.z, o.z) : fminf(acc.z, o.z);
      acc.w = IS_MAX ? fmaxf(acc.w, o.w) : fminf(acc.w, o.w);
    }
    for (int b = 1; b < B; ++b) {
      const float4 v = reinterpret_cast<const float4*>(maps + (size_t)b * n)[i];
      acc.x = IS_MAX ? fmaxf(acc.x, v.x) : fminf(acc.x, v.x);
      acc.y = IS_MAX ? fmaxf(acc.y, v.y) : fminf(acc.y, v.y);
      acc.z = IS_MAX ? fmaxf(acc.z, v.z) : fminf(acc.z, v.z);
      acc.w = IS_MAX ? fmaxf(acc.w, v.w) : fminf(acc.w, v.w);
    }
    reinterpret_cast<float4*>(out)[i] = acc;
  }
  // scalar remainder (everything when rows are not 16-byte aligned)
  for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float acc = maps[i];
    if (accumulate) acc = IS_MAX ? fmaxf(acc, out[i]) : fminf(acc, out[i]);
    for (int b = 1; b < B; ++b)
      acc = IS_MAX ? fmaxf(acc, maps[(size_t)b * n + i]) : fminf(acc, maps[(size_t)b * n + i]);
    out[i] = acc;
  }
}

hipError_t run_fuse_batch(const float* maps, int B, size_t n, float* out, bool is_max,
                          int accumulate, hipStream_t s) {
  // float4 path needs every row 16-byte aligned: n % 4 == 0 and aligned bases
  const bool aligned = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(maps) |
                                        reinterpret_cast<uintptr_t>(out)) % 16 == 0);
  const size_t nvec = aligned ? n : 0;
  const int blocks = blocks_for(aligned ? n / 4 : n, 256, 8192);
  if (is_max)
    hipLaunchKernelGGL(k_fuse_batch<true>, dim3(blocks), dim3(256), 0, s, maps, B, n, nvec, out,
                       accumulate);
  else
    hipLaunchKernelGGL(k_fuse_batch<false>, dim3(blocks), dim3(256), 0, s, maps, B, n, nvec, out,
                       accumulate);
  return hipGetLastError();
}

hipError_t run_mask_from_map(const float* map, float fill, uint8_t* mask, size_t n,
                             hipStream_t s) {
  return launch_finalize(const_cast<float*>(map), nullptr, fill, mask, n, s);
}

}  // namespace dm

// ---------------------------------------------------------------------------
// dm_debug_count_escapes: the silent-drop bound of the LDS-windowed paths, checked ON THE DEVICE.
// Every pixel is projected with the device's own float32 arithmetic (project_pixel); a pixel
// that lands in the map but outside the window -- or the per-row cover -- of the image part that
// owns it would be lost by those paths without a trace.  counts[0] = pixels landing in the map,
// counts[1] = those outside their part's window, counts[2] = those outside their strip's cover.
namespace dm {
namespace {

__global__ void __launch_bounds__(256)
k_count_escapes(View v, int dc, int valid_c, const dm_frame* __restrict__ frames,
                const float* __restrict__ depth, const uint8_t* __restrict__ valid,
                const int* __restrict__ windows, int pc, int pr, int wp, int hp,
                const uint32_t* __restrict__ covers, unsigned long long* __restrict__ counts) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int N = v.H * v.W;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const Cam cam = load_cam(frames + b);
  const int r = i / v.W, q = i - r * v.W;
  bool ok = border_ok(v, r, q);
  if (valid) ok = ok && valid[((size_t)b * valid_c + (valid_c == 1 ? 0 : ch)) * N + i] != 0;
  const float z = depth[((size_t)b * dc + ch) * N + i];
  const Hit h = project_pixel(v, cam, z, ray_x(v, q), ray_y(v, r), ok);
  if (h.cell < 0) return;
  const int zb = h.cell / v.mw, xb = h.cell - zb * v.mw;
  const int part = (r / hp) * pc + q / wp;
  const int* w = windows + ((size_t)b * pc * pr + part) * 4;
  atomicAdd(counts + 0, 1ull);
  if (!(xb >= w[0] && xb < w[0] + w[2] && zb >= w[1] && zb < w[1] + w[3])) atomicAdd(counts + 1, 1ull);
  if (covers) {
    const uint32_t cv = covers[(((size_t)b * v.mh + zb) * pc + part) * 2];
    if (!(xb >= (int)(cv & 0xffffu) && xb < (int)(cv >> 16))) atomicAdd(counts + 2, 1ull);
  }
}

}  // namespace
}  // namespace dm

extern "C" __attribute__((visibility("default"))) int dm_debug_count_escapes(
    const dm_params* p, const dm_frame* frames_host, const float* depth_dev, const uint8_t* valid_dev,
    const int32_t* windows_dev, int32_t pc, int32_t pr, int32_t wp, int32_t hp,
    const uint32_t* covers_dev, unsigned long long* counts_dev, void* workspace_dev, void* stream) {
  using namespace dm;
  if (!p || !frames_host || !depth_dev || !windows_dev || !counts_dev || !workspace_dev || p->B < 1 ||
      pc < 1 || pr < 1 || wp < 1 || hp < 1 || (covers_dev && pr != 1))
    return -1;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (upload_frames(*p, frames_host, workspace_dev, s) != hipSuccess) return -2;
  if (hipMemsetAsync(counts_dev, 0, 3 * sizeof(unsigned long long), s) != hipSuccess) return -2;
  const View v = make_view(*p);
  const int N = p->H * p->W;
  hipLaunchKernelGGL(k_count_escapes, dim3((N + 255) / 256, p->dc, p->B), dim3(256), 0, s, v, p->dc,
                     p->valid_c, static_cast<const dm_frame*>(workspace_dev), depth_dev, valid_dev,
                     windows_dev, pc, pr, wp, hp, covers_dev, counts_dev);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
