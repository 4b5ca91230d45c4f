// LDS-windowed fast path of orth_project (max / min, one value per pixel).
//
// Why: the scatter has no locality the memory system can exploit -- global
// atomics run at ~27 G/s on MI355X whatever their scope (profiles/r01_microbench.log),
// LDS atomics at ~5300 G/s.  So every reduction happens in LDS and each output
// byte is written exactly once, by plain coalesced stores.
//
//   k_window_scatter  one workgroup per (frame, channel, image part).  A part
//       is a column strip (x row band) of the depth image.  The host bounds the
//       part's footprint in the map -- the frustum slab of its pixel rectangle
//       between trunc_depth_min and trunc_depth_max is a convex polytope whose
//       extreme cells are reached at its 8 corners -- and the workgroup
//       accumulates into an LDS image of exactly that window with ds_max_f32 /
//       ds_min_f32, then flushes the window as a "slab" (plain 16-byte stores,
//       read back by the next kernel).  Interleaved with the projection it also
//       writes its share of the map rows outside the frame's union window
//       (fill value, mask 0): ~77 % of the output never sees a second kernel.
//   k_window_merge    one thread per 4 cells of the union window: combines the
//       <= P slabs covering them (windows of neighbouring strips overlap) and
//       writes map and mask once.
//   k_fuse_unions / k_fuse_windows  batch fuse (max / min over the frames), from
//       the finished maps or straight from the slabs.
//
// Everything a workgroup needs is wave-uniform (frame record, windows: one staged
// table per launch) and read through scalar loads; depth is read with 16-byte
// loads, one row segment per group of lanes, several rows in flight per thread,
// the first of them requested before anything but the kernel arguments is known.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <type_traits>
#include <vector>

#include "dm_kernels.hpp"

namespace dm {

namespace {

constexpr int kScatterThreads = 1024;
constexpr int kRowsInFlight = 4;
constexpr int kFillPerHalf = 2;       // fill steps per pipeline half-iteration
constexpr int kMaxLdsBytes = 160 * 1024;


struct Parts {
  int pc, pr;          // column strips x row bands
  int wp, hp;          // part width (multiple of 4) / height in pixels
  int pd;              // depth bands: band k keeps the pixels with depth in [lo_k, hi_k] (every
                       // band reads the whole rectangle; its window is pd times shorter)
};

// Window of one part in map cells; w == 0: the part cannot hit the map.
struct Window {
  int x0, z0, w, h;
};

// Tables of the scatter kernel, one per chunk of frames (= launch), staged into the
// workspace by ONE hipMemcpyAsync per call.  Every access is a scalar load of
// wave-uniform data.  (Passing them as kernel arguments instead saves the copy but
// costs as much at the head of the kernel, and the runtime's kernel-argument pool then
// stalls the host every few dozen launches.)
struct FrameRec {        // what the scatter kernel reads of a dm_frame
  float p[9];            // pitch rotation (row-major)
  float cam_h;
  float y[9];            // yaw rotation (identity for a local map)
  float tx, tz, wo, ho;
  float pad;
};
struct alignas(8) Win16 { short x0, z0, w, h; };     // map sides <= 32767 (8-byte aligned: one scalar load)

constexpr int kChunkFrames = 64;          // frames per scatter launch
constexpr int kChunkWins = 2048;          // part windows per scatter launch
constexpr int kMaxParts = 256;           // parts of a frame (image parts x depth bands)
constexpr int kFewParts = 8;              // window-table row stride for frames of up to 8 parts

struct ScatterTables {
  FrameRec frames[kChunkFrames];
  Win16 unions[kChunkFrames];             // bounding box of a frame's windows, x aligned to 4
  // (frame, part): row stride kFewParts when a frame has at most kFewParts parts (a
  // workgroup then finds its window without first loading the part count), else nparts.
  // Last, so that a call of one chunk copies only the rows it uses.
  Win16 wins[kChunkWins];
};

__device__ __host__ inline Window widen(Win16 w) { return Window{w.x0, w.z0, w.w, w.h}; }
__device__ inline Win16 narrow16(Window w) {
  return Win16{(short)w.x0, (short)w.z0, (short)w.w, (short)w.h};
}
__host__ inline Win16 narrow(Window w) {
  return Win16{(short)w.x0, (short)w.z0, (short)w.w, (short)w.h};
}

// Depth band k of pd over [dmin, dmax]: the same float expressions on the host (window
// bounds) and in the kernel (depth test); neighbouring bands share their boundary value,
// which is harmless for max / min.
__host__ __device__ inline void band_bounds(float dmin, float dmax, int pd, int k, float& lo,
                                            float& hi) {
  const float step = (dmax - dmin) / (float)pd;
  lo = k == 0 ? dmin : __builtin_fmaf((float)k, step, dmin);
  hi = k == pd - 1 ? dmax : __builtin_fmaf((float)(k + 1), step, dmin);
}

__host__ Parts choose_parts(const dm_params& p, int min_parts = 1, int pd = 1) {
  // enough workgroups to fill 256 CUs (and at least min_parts, so that windows fit
  // in LDS), parts not smaller than 32 columns, strip boundaries on 128-byte lines
  // (32 floats) when W allows it
  // ... but not more than pays: the scatter kernel of a small batch is latency bound
  // (~6 us + 0.3 us per pixel and thread), the merge kernel visits every part per cell
  // (~0.37 us per part), so the sum is smallest near sqrt(0.8 * pixels / 1024) parts
  // (measured at B = 1, 320x240: 80 parts 43 us, 10 parts 15 us)
  const long frames = (long)p.B * (p.vc ? p.vc : p.dc);
  int want = (int)((256 + frames - 1) / frames);
  const int pays = (int)lround(sqrt(0.8 * (double)p.H * p.W / 1024.0));
  if (want > pays) want = pays;
  want = (want + pd - 1) / pd;                 // image parts: the depth bands multiply them
  if (want < min_parts) want = min_parts;
  if (want < 1) want = 1;
  Parts s;
  s.pc = 1; s.pr = 1; s.pd = pd;
  const int unit = (p.W % 32 == 0) ? 32 : 4;
  const int units = (p.W + unit - 1) / unit;
  int pc = want < units ? want : units;
  if (pc > 16) pc = 16;
  // prefer a divisor of the unit count (equal strips)
  while (pc > 1 && units % pc != 0) --pc;
  s.pc = pc;
  s.wp = ((units + pc - 1) / pc) * unit;
  int pr = (want + pc - 1) / pc;
  if (pr > 8) pr = 8;
  if (pr > p.H / 16) pr = p.H / 16 > 0 ? p.H / 16 : 1;
  s.pr = pr;
  s.hp = (p.H + pr - 1) / pr;
  return s;
}

// y = RN(1/b) in float32, exactly: pick the neighbour minimising |b*y - 1|
// (b*y is exact in double).
__host__ bool exact_reciprocal(float b, float* y_out) {
  if (!(b > 0.0f) || !isfinite(b)) return false;
  float y = (float)(1.0 / (double)b);
  if (!isfinite(y) || y < 1e-30f || y > 1e30f) return false;
  // the Markstein step needs b's reciprocal rounded to nearest
  float best = y;
  double err = fabs(fma((double)b, (double)y, -1.0));
  const float cand[2] = {nextafterf(y, 0.0f), nextafterf(y, INFINITY)};
  for (float c : cand) {
    const double e = fabs(fma((double)b, (double)c, -1.0));
    if (e < err) { err = e; best = c; }
  }
  *y_out = best;
  return true;
}

__host__ bool axis_aligned(const FrameRec* f, int B) {
  for (int b = 0; b < B; ++b) {
    const float* p = f[b].p; const float* y = f[b].y;
    if (!(p[0] == 1.0f && p[1] == 0.0f && p[2] == 0.0f && p[3] == 0.0f && p[6] == 0.0f))
      return false;
    if (!(y[1] == 0.0f && y[3] == 0.0f && y[4] == 1.0f && y[5] == 0.0f && y[7] == 0.0f))
      return false;
  }
  return true;
}

// Cell coordinates are affine in (ax*z, ay*z, z) for a given frame:
//   xf = z * (xa*ax + xb*ay + xc) + xd,   zf = z * (za*ax + zb*ay + zc) + zd
// (ax, ay = ray slopes of the pixel).  One coefficient set per frame, in double.
struct FrameAffine {
  double xa, xb, xc, xd, za, zb, zc, zd;
  bool finite;
};

__host__ FrameAffine frame_affine(const dm_params& p, const dm_frame& f) {
  // local = Rp^T-chain(X, Y, Z) + (0, h, 0);  global = Ry-chain(local) + (tx, 0, tz)
  double L[3][3], t1[3] = {0.0, f.cam_height, 0.0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) L[i][j] = f.Rp[3 * j + i];           // out_i = sum_j R[j][i] p_j
  double G[3][3], t2[3];
  if (p.to_global) {
    double Y[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Y[i][j] = f.Ry[3 * j + i];
    const double tr[3] = {f.tx, 0.0, f.tz};
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) G[i][j] = Y[i][0] * L[0][j] + Y[i][1] * L[1][j] + Y[i][2] * L[2][j];
      t2[i] = Y[i][0] * t1[0] + Y[i][1] * t1[1] + Y[i][2] * t1[2] + tr[i];
    }
  } else {
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) G[i][j] = L[i][j]; t2[i] = t1[i]; }
  }
  FrameAffine a;
  const double inv = 1.0 / p.res;
  a.xa = G[0][0] * inv; a.xb = G[0][1] * inv; a.xc = G[0][2] * inv;
  a.xd = t2[0] * inv + f.width_offset;
  double za = G[2][0] * inv, zb = G[2][1] * inv, zc = G[2][2] * inv;
  double zd = t2[2] * inv + f.height_offset;
  if (p.flip_h) { za = -za; zb = -zb; zc = -zc; zd = (double)(p.mh - 1) - zd; }
  a.za = za; a.zb = zb; a.zc = zc; a.zd = zd;
  a.finite = isfinite(a.xa) && isfinite(a.xb) && isfinite(a.xc) && isfinite(a.xd) &&
             isfinite(a.za) && isfinite(a.zb) && isfinite(a.zc) && isfinite(a.zd);
  return a;
}

// Footprint of the pixel rectangle [q0,q1) x [r0,r1) of a frame in map cells,
// padded by 2 cells and aligned to 4 columns, clipped to the map.  The ray slopes of the
// rectangle's border pixels do not depend on the frame: PartSlopes holds them per part.
struct PartSlopes {
  double ax[2], ay[2];        // (q - cx) / fx at q0, q1 - 1;  (y - cy) / fy at r0, r1 - 1
  bool empty;                 // nothing left of the part after clip_border
};

__host__ PartSlopes part_slopes(const dm_params& p, int q0, int q1, int r0, int r1) {
  PartSlopes s;
  if (p.clip_border > 0) {
    const int c = p.clip_border;
    if (q0 < c) q0 = c;
    if (r0 < c) r0 = c;
    if (q1 > p.W - c) q1 = p.W - c;
    if (r1 > p.H - c) r1 = p.H - c;
  }
  s.empty = q0 >= q1 || r0 >= r1;
  const int qs[2] = {q0, q1 - 1}, rs[2] = {r0, r1 - 1};
  for (int i = 0; i < 2; ++i) {
    s.ax[i] = ((double)qs[i] - p.cx) / p.fx;
    double yr = rs[i];
    if (p.flip_h) yr = (double)(p.H - 1) - yr;
    s.ay[i] = (yr - p.cy) / p.fy;
  }
  return s;
}

__host__ bool frustum_bounded(const dm_params& p) {
  return p.has_dmin && p.has_dmax && p.dmin >= 0.0f && p.dmax >= p.dmin && isfinite(p.dmax);
}

__host__ Window part_window(const dm_params& p, const FrameAffine& fa, const PartSlopes& ps,
                            bool bounded, float dlo, float dhi) {
  if (ps.empty) return Window{0, 0, 0, 0};
  if (!bounded || !fa.finite) return Window{0, 0, p.mw, p.mh};
  double lo_x = INFINITY, hi_x = -INFINITY, lo_z = INFINITY, hi_z = -INFINITY;
  double poison = 0.0;                         // NaN as soon as one corner is not finite
  const double zs[2] = {dlo, dhi};
  for (int qi = 0; qi < 2; ++qi)
    for (int ri = 0; ri < 2; ++ri) {
      const double sx = fa.xa * ps.ax[qi] + fa.xb * ps.ay[ri] + fa.xc;
      const double sz = fa.za * ps.ax[qi] + fa.zb * ps.ay[ri] + fa.zc;
      for (int zi = 0; zi < 2; ++zi) {
        const double xf = zs[zi] * sx + fa.xd, zf = zs[zi] * sz + fa.zd;
        lo_x = xf < lo_x ? xf : lo_x; hi_x = xf > hi_x ? xf : hi_x;
        lo_z = zf < lo_z ? zf : lo_z; hi_z = zf > hi_z ? zf : hi_z;
        poison += xf * 0.0 + zf * 0.0;
      }
    }
  if (!(poison == 0.0)) return Window{0, 0, p.mw, p.mh};
  // cells are floor(v + 0.5); 2 cells of slack cover the float32 rounding of
  // the device arithmetic (observed error < 1e-3 cell)
  double x0 = floor(lo_x + 0.5) - 2, x1 = floor(hi_x + 0.5) + 3;
  double z0 = floor(lo_z + 0.5) - 2, z1 = floor(hi_z + 0.5) + 3;
  if (x0 < 0) x0 = 0;
  if (z0 < 0) z0 = 0;
  if (x1 > p.mw) x1 = p.mw;
  if (z1 > p.mh) z1 = p.mh;
  if (x0 >= x1 || z0 >= z1) return Window{0, 0, 0, 0};
  Window w;
  w.x0 = ((int)x0) & ~3;
  const int xe = ((int)x1 + 3) & ~3;           // mw % 4 == 0 is a precondition
  w.w = (xe > p.mw ? p.mw : xe) - w.x0;
  w.z0 = (int)z0;
  w.h = (int)z1 - w.z0;
  return w;
}

// ---------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------
// Scalars of the scatter kernel, slimmed down to what it reads (SGPR budget).
// Disabled tests are encoded as always-true bounds (dmin = -inf, dmax = hmax =
// +inf: they then only reject NaN, which never reaches the map anyway) and a
// local-space projection as an identity yaw with zero translation, so the
// pixel loop has no flag to branch on.
// -DDM_STAMPS: instrumented build for tools/phase_stamps.py -- thread 0 of every
// workgroup of k_window_scatter records the 100 MHz real-time counter at phase boundaries.
#ifdef DM_STAMPS
#define DM_STAMP(k) do { long long t_; \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    stamp[k] = t_; } while (0)
#define DM_STAMPS_OUT() do { if (threadIdx.x == 0 && a.stamps) \
    for (int k_ = 0; k_ < 12; ++k_) \
      a.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 12 + k_] = stamp[k_]; \
  } while (0)
static long long* g_stamp_buffer = nullptr;
#else
#define DM_STAMP(k) do { } while (0)
#define DM_STAMPS_OUT() do { } while (0)
#endif

struct ScatterArgs {
  int W, H;
  int clip;                   // border pixels to drop (0 = none)
  int flip_h;
  float cx, cy, fx, fy, res;
  float fx_inv, fy_inv, res_inv;
  float dmin, dmax, hmax;
  float Hm1, mhm1;
  Parts parts;
  int dc, valid_c;
  int oc;                     // output channels per frame handled by this pass
  int ch0;                    // first output channel of this launch (channel groups)
  int oc_total;               // channels of `out` / `value`
  int slab_stride;            // cells per slab
  float fill;
#ifdef DM_STAMPS
  long long* stamps;
#endif
  int b0;                     // first frame of this launch's chunk
  Win16* g_wins;              // (B, nparts)  device copies for the kernels that follow
  Win16* g_unions;            // (B)
  const float* depth;
  const float* value;         // (B, oc_total, H, W) or NULL: project the heights
  const uint8_t* valid;
  float* slabs;
  // fill duty: the part of every output map outside its frame's union window
  float* out;
  uint8_t* mask;
  int mh, mw;
};

// (int)floorf(x) in one instruction; NaN -> 0, saturating (like v_cvt_i32_f32)
__device__ inline int floor_to_int(float x) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// global load and store of the wave (s_waitcnt vmcnt(0)), which would expose the latency
// of the depth rows and fill stores deliberately left in flight across it.
__device__ inline void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ds_max_f32 / ds_min_f32: "store if new > old" -- torch_scatter's rule; a NaN
// operand never replaces a number.
template <bool IS_MAX>
__device__ inline void lds_reduce(float* cell, float v) {
  if (IS_MAX) __hip_atomic_fetch_max(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_min(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// FAST: every frame's rotations have the exact 0/1 pattern of rotate([1,0,0],.)
//       and rotate([0,1,0],.) AND the Markstein reciprocals are usable
//       (dm_pixel.hpp).  !FAST: full FMA chains and IEEE division.
// VEC = 4: 16-byte depth loads (W % 4 == 0, 16-byte aligned base); VEC = 1: any shape.
// HAS_VALUE: scatter value[b, ch] (maps.py:314-316) instead of the height.
// LEAN: finite depth bounds on both sides, no height truncation, no poisoned rays
//       (no border clip, no valid map): a non-finite or out-of-range pixel is then
//       already rejected by the two depth compares, so the ordered-compare and the
//       height compare are dropped.  (Pipeline-tail rows are poisoned through z.)
template <bool IS_MAX, bool FAST, bool HAS_VALID, bool HAS_VALUE, int VEC, bool LEAN>
__global__ void __launch_bounds__(kScatterThreads)
k_window_scatter(ScatterArgs a, const ScatterTables* __restrict__ tables) {
  const ScatterTables& t = *tables;            // this launch's chunk of frames
  extern __shared__ float lds[];
  const int part = blockIdx.x;                 // pr-major, pc-minor
  const int chl = blockIdx.y;                  // channel within this launch's group
  const int bl = blockIdx.z, b = a.b0 + bl;    // frame within the chunk / in the batch
  const int ch = a.ch0 + chl;                  // output channel
  const int dch = a.dc == 1 ? 0 : ch;          // depth / cell-index channel (utils.py:475-477)
  const int nparts = a.parts.pc * a.parts.pr * a.parts.pd;
  const int pcx = part % a.parts.pc, pry = (part / a.parts.pc) % a.parts.pr;
  const int pdk = part / (a.parts.pc * a.parts.pr);          // depth band
  // The part's pixel rectangle and this thread's place in it need nothing but kernel
  // arguments, so the first depth rows are requested before anything of the staged
  // table has arrived (window, union window, frame record: all of the head of the kernel
  // runs under these loads).
  const int q0 = pcx * a.parts.wp;
  int q1 = q0 + a.parts.wp; if (q1 > a.W) q1 = a.W;
  const int r0 = pry * a.parts.hp;
  int r1 = r0 + a.parts.hp; if (r1 > a.H) r1 = a.H;
  const int nx = (q1 - q0 + VEC - 1) / VEC;    // lane groups per row
  const int ntx = nx < kScatterThreads ? nx : kScatterThreads;
  const int rows_per_iter = kScatterThreads / ntx;
  const int gx = threadIdx.x % ntx, gy = threadIdx.x / ntx;
  const size_t N = (size_t)a.H * a.W;
  const float* dimg = a.depth + ((size_t)b * a.dc + dch) * N;
  const uint8_t* vimg = HAS_VALID
      ? a.valid + ((size_t)b * a.valid_c + (a.valid_c == 1 ? 0 : dch)) * N : nullptr;
  const float* simg = HAS_VALUE ? a.value + ((size_t)b * a.oc_total + ch) * N : nullptr;
  const float qnan = __builtin_nanf("");
  float za[kRowsInFlight][VEC], zb_[kRowsInFlight][VEC];
  float va[HAS_VALUE ? kRowsInFlight : 1][VEC], vb_[HAS_VALUE ? kRowsInFlight : 1][VEC];
  auto load_rows_at = [&](float (&z)[kRowsInFlight][VEC],
                          float (&sv)[HAS_VALUE ? kRowsInFlight : 1][VEC], int q, int r) {
#pragma unroll
    for (int u = 0; u < kRowsInFlight; ++u) {
      // rows past the part are clamped to its last row (a legal address);
      // project_rows() ignores them.  No per-lane branch: the loads stay in
      // one basic block and the compiler can wait on them individually.
      int rr = r + u * rows_per_iter;
      rr = rr < r1 ? rr : r1 - 1;
      if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(dimg + (size_t)rr * a.W + q);
        z[u][0] = t.x; z[u][1 % VEC] = t.y; z[u][2 % VEC] = t.z; z[u][3 % VEC] = t.w;
      } else {
        z[u][0] = dimg[(size_t)rr * a.W + q];
      }
      if (HAS_VALID) {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          z[u][k] = vimg[(size_t)rr * a.W + q + k] ? z[u][k] : qnan;
      }
      if (HAS_VALUE) {
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(simg + (size_t)rr * a.W + q);
          sv[u][0] = t.x; sv[u][1 % VEC] = t.y; sv[u][2 % VEC] = t.z; sv[u][3 % VEC] = t.w;
        } else {
          sv[u][0] = simg[(size_t)rr * a.W + q];
        }
      }
    }
  };
  bool first_rows_loaded = false;
  if (gx < nx && q1 > q0 && r1 > r0) {         // (always, for a non-empty part)
    load_rows_at(za, va, q0 + gx * VEC, r0 + gy);
    first_rows_loaded = true;
  }

  // Everything this workgroup reads from the staged table -- its window, the frame's union
  // window and the frame record -- is requested in ONE batch of scalar loads and pinned
  // (the asm makes the values opaque): left to the compiler these loads trickle in close
  // to their first use, one dependent ~0.7-us round trip after the other.
  int w_raw[2], u_raw[2];
  float fr[23];
  {
    const int* tw = reinterpret_cast<const int*>(&t.wins[bl * kFewParts + (part & (kFewParts - 1))]);
    const int* tu = reinterpret_cast<const int*>(&t.unions[bl]);
    const float* tf = reinterpret_cast<const float*>(&t.frames[bl]);
    w_raw[0] = tw[0]; w_raw[1] = tw[1]; u_raw[0] = tu[0]; u_raw[1] = tu[1];
#pragma unroll
    for (int i = 0; i < 23; ++i) fr[i] = tf[i];
    asm volatile("" : "+s"(w_raw[0]), "+s"(w_raw[1]), "+s"(u_raw[0]), "+s"(u_raw[1]),
                      "+s"(fr[0]), "+s"(fr[1]), "+s"(fr[2]), "+s"(fr[3]), "+s"(fr[4]), "+s"(fr[5]),
                      "+s"(fr[6]), "+s"(fr[7]), "+s"(fr[8]), "+s"(fr[9]), "+s"(fr[10]), "+s"(fr[11]),
                      "+s"(fr[12]), "+s"(fr[13]), "+s"(fr[14]), "+s"(fr[15]), "+s"(fr[16]),
                      "+s"(fr[17]), "+s"(fr[18]), "+s"(fr[19]), "+s"(fr[20]), "+s"(fr[21]),
                      "+s"(fr[22]));
  }
  const Win16 w_few = {(short)(w_raw[0] & 0xffff), (short)(w_raw[0] >> 16),
                       (short)(w_raw[1] & 0xffff), (short)(w_raw[1] >> 16)};
  const Window w = nparts <= kFewParts ? widen(w_few) : widen(t.wins[bl * nparts + part]);
  const int area = w.w * w.h;                  // 0: nothing of this part can land
#ifdef DM_STAMPS
  long long stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DM_STAMP(0);
  // younger waves of a SIMD get the higher issue priority (age arbitration favours the
  // oldest wave otherwise, and the last wave left on a SIMD runs latency bound)
  {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave >= 12) __builtin_amdgcn_s_setprio(3);
    else if (wave >= 8) __builtin_amdgcn_s_setprio(2);
    else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  }

  // Fill duty, interleaved with the scatter so that these stores ride under the
  // projection: map rows part, part + nparts, ... of (b, ch), minus the frame's union
  // window U (k_window_merge writes U).  One float4 (+ 4 mask bytes) per thread and step,
  // ALWAYS executed: an element that needs no store (inside U, or past the end) is
  // redirected to `alt`, a cell of this workgroup's share that does get the fill value,
  // so the stores are unconditional straight-line code and the compiler can count the
  // pipelined loop's waits exactly.
  const Window U = {(short)(u_raw[0] & 0xffff), (short)(u_raw[0] >> 16),
                    (short)(u_raw[1] & 0xffff), (short)(u_raw[1] >> 16)};
  const int g4 = a.mw >> 2;
  const int fill_rows = (a.mh - part + nparts - 1) / nparts;
  const size_t map_base = ((size_t)b * a.oc_total + ch) * (size_t)a.mh * a.mw;
  // where redirected stores go: a cell of this share outside U if there is one (it gets
  // the fill value anyway), else the share's first cell (inside U: k_window_merge,
  // which runs after this kernel, overwrites it)
  int alt_cell = part * a.mw;
  if (fill_rows > 0 && U.w > 0 && U.h > 0 && U.x0 == 0) {
    const int last_row = part + (fill_rows - 1) * nparts;
    if (U.x0 + U.w < a.mw) alt_cell = part * a.mw + U.x0 + U.w;                  // right of U
    else if (part >= U.z0 && last_row >= U.z0 + U.h) alt_cell = last_row * a.mw; // below U
  }
  const bool do_fill = a.out != nullptr && fill_rows > 0;                       // wave-uniform
  const int fill_total = do_fill ? fill_rows * g4 : 0;
  const int fill_steps = (fill_total + kScatterThreads - 1) / kScatterThreads;
  const float g4_inv = 1.0f / (float)g4;
  int fs = 0;
  auto fill_step = [&]() {
    const int i = fs * kScatterThreads + (int)threadIdx.x;     // < 2^24
    ++fs;
    int k = (int)((float)i * g4_inv);          // i / g4 without a division routine or a branch
    k -= (k * g4 > i);
    k += ((k + 1) * g4 <= i);
    const int g = i - k * g4;
    const int r = part + k * nparts, x = g << 2;
    const bool skip = i >= fill_total || ((unsigned)(r - U.z0) < (unsigned)U.h &&
                                          (unsigned)(x - U.x0) < (unsigned)U.w);
    int cell = r * a.mw + x;
    asm("" : "+v"(cell));                      // keep the select a v_cndmask
    cell = skip ? alt_cell : cell;
    *reinterpret_cast<float4*>(a.out + map_base + cell) = make_float4(a.fill, a.fill, a.fill, a.fill);
    *reinterpret_cast<uint32_t*>(a.mask + map_base + cell) = 0u;
  };
  // device copies of the geometry for the kernels that follow (stored at the very end: a
  // store in flight makes its wave wait before the first depth loads)
  auto publish_geometry = [&]() {
    if (chl == 0 && threadIdx.x == 0) {
      a.g_wins[(size_t)b * nparts + part] = narrow16(w);
      if (part == 0) a.g_unions[b] = narrow16(U);
    }
  };
  if (area == 0) {                             // wave-uniform
    while (fs < fill_steps) fill_step();
    publish_geometry();
    return;
  }
  // FrameRec: p[9], cam_h, y[9], tx, tz, wo, ho
  // pitch: rows 1,2 of R; yaw: rows 0,2 (the rest is 0/1 when FAST)
  const float p0 = fr[0], p1 = fr[1], p2 = fr[2], p3 = fr[3], p4 = fr[4],
              p5 = fr[5], p6 = fr[6], p7 = fr[7], p8 = fr[8];
  const float y0 = fr[10], y1r = fr[11], y2r = fr[12], y3 = fr[13], y4 = fr[14],
              y5 = fr[15], y6 = fr[16], y7 = fr[17], y8 = fr[18];
  const float cam_h = fr[9], tx = fr[19], tz = fr[20];
  const float wo = fr[21], ho = fr[22];
  const float flip_s = a.flip_h ? -1.0f : 1.0f, flip_c = a.flip_h ? a.mhm1 : 0.0f;
  DM_STAMP(1);
  bool lds_ready = false;
  const unsigned dummy = (unsigned)area + (threadIdx.x & 63u);   // 64 scratch cells after the window
  float band_lo = a.dmin, band_hi = a.dmax;    // this part's depth band (wave-uniform)
  if (a.parts.pd > 1) band_bounds(a.dmin, a.dmax, a.parts.pd, pdk, band_lo, band_hi);

  {
    for (int g = gx; g < nx; g += ntx) {       // one trip unless the strip is wider than the block
      const int q = q0 + g * VEC;
      // ray slope of each column (maps.py:677); border columns are poisoned with
      // NaN, which flows through X to the cell coordinates (maps.py:48-70)
      float ax[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float d = (float)(q + k) - a.cx;
        ax[k] = FAST ? div_markstein(d, a.fx, a.fx_inv) : d / a.fx;
        // (idle threads and pipeline-tail rows simply repeat the part's last row: a max /
        // min reduction is idempotent, so they need no poison)
        if (!LEAN) ax[k] = (q + k < a.clip || q + k >= a.W - a.clip) ? qnan : ax[k];
      }
      // Software pipeline over groups of kRowsInFlight rows: the loads of group
      // i+1 are in flight while group i is projected (all waves of a workgroup
      // run in phase, so latency has to be hidden inside each wave).
      const int step = rows_per_iter * kRowsInFlight;
      auto load_rows = [&](float (&z)[kRowsInFlight][VEC],
                           float (&sv)[HAS_VALUE ? kRowsInFlight : 1][VEC], int r) {
        load_rows_at(z, sv, q, r);
      };
      auto project_rows = [&](const float (&z)[kRowsInFlight][VEC],
                              const float (&sv)[HAS_VALUE ? kRowsInFlight : 1][VEC], int r) {
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
          int rr = r + u * rows_per_iter;
          rr = rr < r1 ? rr : r1 - 1;                            // tail: repeat the last row
          float yr = (float)rr;
          yr = a.flip_h ? a.Hm1 - yr : yr;                       // maps.py:670-671
          const float dy = yr - a.cy;
          float ay = FAST ? div_markstein(dy, a.fy, a.fy_inv) : dy / a.fy;
          // rows in the clipped border: poison
          if (!LEAN) ay = (rr < a.clip || rr >= a.H - a.clip) ? qnan : ay;
          // the VEC pixels of a row are projected side by side (independent
          // chains for the scheduler); their LDS atomics come last so that no
          // branch separates the arithmetic of neighbouring pixels
          unsigned li[VEC];
          float hv[VEC];
          bool ok[VEC];
          float xfv[VEC], zfv[VEC], h1v[VEC], h2v[VEC];
          if (FAST && VEC == 4) {
            // two pixels per instruction (v_pk_mul/fma/add_f32): the kernel is bound by
            // dependent-instruction issue, and the packed forms have the same rounding
            // per element as the scalar ones
            typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int k = 0; k < VEC; k += 2) {
              const f2 zz = {z[u][k], z[u][(k + 1) % VEC]};
              const f2 axp = {ax[k], ax[(k + 1) % VEC]};
              const f2 X = axp * zz;
              const f2 Y = zz * ay;                                            // maps.py:677-678
              f2 h1 = __builtin_elementwise_fma(zz, (f2){p7, p7}, Y * p4) + cam_h;   // maps.py:790-797
              const f2 z1 = __builtin_elementwise_fma(zz, (f2){p8, p8}, Y * p5);
              const f2 x2 = __builtin_elementwise_fma(z1, (f2){y6, y6}, X * y0) + tx;   // maps.py:884-892
              const f2 z2 = __builtin_elementwise_fma(z1, (f2){y8, y8}, X * y2r) + tz;
              const f2 ri = {a.res_inv, a.res_inv}, nres = {-a.res, -a.res};
              const f2 qx = x2 * ri, qz = z2 * ri;                             // exact division
              f2 xf2 = __builtin_elementwise_fma(__builtin_elementwise_fma(nres, qx, x2), ri, qx) + wo;
              f2 zf2 = __builtin_elementwise_fma(__builtin_elementwise_fma(nres, qz, z2), ri, qz) + ho;
              zf2 = __builtin_elementwise_fma(zf2, (f2){flip_s, flip_s}, (f2){flip_c, flip_c});
              xf2 = xf2 + 0.5f;
              zf2 = zf2 + 0.5f;
              xfv[k] = xf2.x; xfv[(k + 1) % VEC] = xf2.y;
              zfv[k] = zf2.x; zfv[(k + 1) % VEC] = zf2.y;
              h1v[k] = h1.x; h1v[(k + 1) % VEC] = h1.y;
              h2v[k] = h1.x; h2v[(k + 1) % VEC] = h1.y;
            }
          } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
              const float zz = z[u][k];
              const float X = ax[k] * zz, Y = ay * zz;             // maps.py:677-678
              float x1, h1, z1, x2, h2, z2;
              if (FAST) {
                x1 = X;
                h1 = __builtin_fmaf(zz, p7, Y * p4) + cam_h;       // maps.py:790-797
                z1 = __builtin_fmaf(zz, p8, Y * p5);
                x2 = __builtin_fmaf(z1, y6, x1 * y0) + tx;         // maps.py:884-892
                z2 = __builtin_fmaf(z1, y8, x1 * y2r) + tz;
                h2 = h1;
              } else {
                x1 = __builtin_fmaf(zz, p6, __builtin_fmaf(Y, p3, X * p0)) + 0.0f;
                h1 = __builtin_fmaf(zz, p7, __builtin_fmaf(Y, p4, X * p1)) + cam_h;
                z1 = __builtin_fmaf(zz, p8, __builtin_fmaf(Y, p5, X * p2)) + 0.0f;
                x2 = __builtin_fmaf(z1, y6, __builtin_fmaf(h1, y3, x1 * y0)) + tx;
                h2 = __builtin_fmaf(z1, y7, __builtin_fmaf(h1, y4, x1 * y1r)) + 0.0f;
                z2 = __builtin_fmaf(z1, y8, __builtin_fmaf(h1, y5, x1 * y2r)) + tz;
              }
              float xf = (FAST ? div_markstein(x2, a.res, a.res_inv) : x2 / a.res) + wo;
              float zf = (FAST ? div_markstein(z2, a.res, a.res_inv) : z2 / a.res) + ho;
              // flip: (mh-1) - zf as fma(zf, -1, mh-1); no flip: fma(zf, 1, 0) -- both exact
              zf = __builtin_fmaf(zf, flip_s, flip_c);             // maps.py:1006-1009
              xfv[k] = xf + 0.5f;                                  // maps.py:1012-1013
              zfv[k] = zf + 0.5f;
              h1v[k] = h1; h2v[k] = h2;
            }
          }
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const float zz = z[u][k], xf = xfv[k], zf = zfv[k], h1 = h1v[k], h2 = h2v[k];
            // floor + convert in one instruction; window test in integers (the
            // window lies inside the map).  The conversion saturates and maps
            // NaN to 0, so NaN is excluded by the ordered compare.
            // maps.py:537-544, 286-288, 1150-1158
            const unsigned ux = (unsigned)(floor_to_int(xf) - w.x0);
            const unsigned uz = (unsigned)(floor_to_int(zf) - w.z0);
            ok[k] = ux < (unsigned)w.w && uz < (unsigned)w.h && zz <= band_hi && zz >= band_lo;
            if (!LEAN) ok[k] = ok[k] && !__builtin_isunordered(xf, zf) && h1 <= a.hmax;
            if (!FAST && !HAS_VALUE) ok[k] = ok[k] && (h2 == h2);
            const float sval = HAS_VALUE ? sv[u][k] : h2;
            if (HAS_VALUE) ok[k] = ok[k] && (sval == sval);      // NaN never replaces a number
            // rejected pixels are redirected to a per-lane dummy cell behind the
            // window instead of being branched around: the whole row group stays
            // one basic block the scheduler can interleave
            unsigned cell = __umul24(uz, (unsigned)w.w) + ux;
            asm("" : "+v"(cell));   // keep the select below a v_cndmask, not a branch
            li[k] = ok[k] ? cell : dummy;
            hv[k] = sval;
          }
          // Neighbouring pixels of a row often share a cell (walls, near floor), and lanes
          // that hit one LDS address serialise: +30 % kernel time on scene-like depth.  If
          // any thread of the wave has its four pixels in ONE cell (wave-uniform test), every
          // run of equal cells inside a thread is reduced in registers and only its last
          // pixel issues the atomic.  (Unconditionally the extra selects cost 2.6 us on
          // incoherent depth; behind the test 1 us.)
          if (VEC == 4 && __builtin_amdgcn_ballot_w64(li[0] == li[VEC - 1] && li[0] != dummy) != 0) {
#pragma unroll
            for (int k = 0; k + 1 < VEC; ++k) {
              const bool same = li[k] == li[k + 1];
              const float m = IS_MAX ? fmaxf(hv[k], hv[k + 1]) : fminf(hv[k], hv[k + 1]);
              hv[k + 1] = same ? m : hv[k + 1];
              li[k] = same ? dummy : li[k];
            }
          }
#pragma unroll
          for (int k = 0; k < VEC; ++k) lds_reduce<IS_MAX>(lds + li[k], hv[k]);
        }
      };
      const int niter = (r1 - r0 + step - 1) / step;             // wave-uniform
      // Two copies of the pipelined loop, with and without fill duty, chosen by ONE
      // wave-uniform branch: inside, exactly kFillPerHalf unconditional fill stores
      // follow each group of loads, so the waits count them and never wait on a store
      // or on the prefetch.
      auto pipeline = [&](auto with_fill) {
        constexpr bool kFill = decltype(with_fill)::value;
        int r = r0 + gy;
        if (!first_rows_loaded) load_rows(za, va, r);      // (later trips of the column loop)
        first_rows_loaded = false;
        DM_STAMP(2);
        // the LDS window is initialised while the first depth rows are in flight
        if (!lds_ready) {                      // wave-uniform, first trip only
          for (int i = threadIdx.x * 4; i < area; i += kScatterThreads * 4)
            *reinterpret_cast<float4*>(lds + i) = make_float4(a.fill, a.fill, a.fill, a.fill);
          lds_barrier();
          lds_ready = true;
        }
        DM_STAMP(3);
        for (int it = 0; it < niter; it += 2) {
          load_rows(zb_, vb_, r + step);
          if (kFill) {
#pragma unroll
            for (int t = 0; t < kFillPerHalf; ++t) fill_step();
          }
          project_rows(za, va, r);
          if (it + 1 < niter) {
            load_rows(za, va, r + 2 * step);
            if (kFill) {
#pragma unroll
              for (int t = 0; t < kFillPerHalf; ++t) fill_step();
            }
            project_rows(zb_, vb_, r + step);
          }
          r += 2 * step;
        }
      };
      if (do_fill) pipeline(std::true_type{}); else pipeline(std::false_type{});
    }
  }
  DM_STAMP(4);
  while (fs < fill_steps) fill_step();
  lds_barrier();
  DM_STAMP(5);
  const int pid = (b * a.oc + chl) * nparts + part;      // slabs are per channel group
  float* slab = a.slabs + (size_t)pid * a.slab_stride;
  for (int i = threadIdx.x * 4; i < area; i += kScatterThreads * 4)
    *reinterpret_cast<float4*>(slab + i) = *reinterpret_cast<const float4*>(lds + i);
  publish_geometry();
  DM_STAMP(6);
  DM_STAMPS_OUT();
}

// Table ring: the first thread of a kernel that follows k_window_scatter in stream order tells
// the host that the scatter's table slot may be rewritten (a word in pinned host memory).
// Called on the way OUT of the kernel: a store ahead of the loads of the read-only window
// tables (even an opaque one) turns them from scalar into vector loads, each waited for in
// turn (k_window_merge: 12.2 instead of 9.4 us).
__device__ inline void signal_slot_free(uint32_t* signal, uint32_t ticket) {
  if (signal && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(signal), "v"(ticket));
}

struct MergeArgs {
  int b0, oc, ch0, oc_total, mh, mw;  // oc channels per frame in this launch, starting at ch0
  int nparts;                 // pc * pr
  int slab_stride;
  float fill;
  // this chunk's part windows (row stride win_stride) and union windows: in the staged
  // table (already in every XCD's L2) or, when that is a slot of the table ring, the device
  // copies k_window_scatter leaves behind
  const Win16* wins;
  const Win16* unions;
  int win_stride;
  uint32_t* signal;           // table ring: k_window_scatter of this stream position is done
  uint32_t ticket;
  const float* slabs;
  float* out;
  uint8_t* mask;
};

constexpr int kMergeThreads = 256;

// Writes the union window U of every (frame, channel): max/min over the slabs covering
// each cell, fill where none does.  One float4 group per thread, row-major inside U, so
// a wave's store covers up to 1 KiB (map) / 256 B (mask) contiguously.
template <bool IS_MAX>
__global__ void __launch_bounds__(kMergeThreads)
k_window_merge(MergeArgs a) {
  const int fcl = blockIdx.y;                  // (frame in chunk) * oc + channel of the group
  const int bl = fcl / a.oc, b = a.b0 + bl;
  const int fc = fcl + a.b0 * a.oc;            // slab index of (frame, channel)
  const size_t fo = (size_t)b * a.oc_total + a.ch0 + (fcl - bl * a.oc);   // map index in `out`
  const Window U = widen(a.unions[bl]);
  const int win_stride = a.win_stride;
  const int ug4 = U.w >> 2;                    // float4 groups per U row
  const int total = ug4 * U.h;
  const int i = blockIdx.x * kMergeThreads + threadIdx.x;
  if (i >= total) {
    signal_slot_free(a.signal, a.ticket);
    return;
  }
  const int row = i / ug4;
  const int zb = U.z0 + row, x = U.x0 + ((i - row * ug4) << 2);
  float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
  for (int p = 0; p < a.nparts; ++p) {
    const Window w = widen(a.wins[bl * win_stride + p]);
    if (w.w == 0) continue;
    const unsigned ux = (unsigned)(x - w.x0), uz = (unsigned)(zb - w.z0);
    if (ux >= (unsigned)w.w || uz >= (unsigned)w.h) continue;
    const float* slab = a.slabs + ((size_t)fc * a.nparts + p) * a.slab_stride;
    const float4 s = *reinterpret_cast<const float4*>(slab + (size_t)uz * w.w + ux);
    acc.x = IS_MAX ? fmaxf(acc.x, s.x) : fminf(acc.x, s.x);
    acc.y = IS_MAX ? fmaxf(acc.y, s.y) : fminf(acc.y, s.y);
    acc.z = IS_MAX ? fmaxf(acc.z, s.z) : fminf(acc.z, s.z);
    acc.w = IS_MAX ? fmaxf(acc.w, s.w) : fminf(acc.w, s.w);
  }
  const size_t cell = fo * (size_t)a.mh * a.mw + (size_t)zb * a.mw + x;
  *reinterpret_cast<float4*>(a.out + cell) = acc;
  const uint32_t mk = (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
                      ((uint32_t)mask_of(acc.z, a.fill) << 16) |
                      ((uint32_t)mask_of(acc.w, a.fill) << 24);
  *reinterpret_cast<uint32_t*>(a.mask + cell) = mk;
  signal_slot_free(a.signal, a.ticket);
}

// The same for frames of many parts (image parts x depth bands): a block owns a tile of
// kTileGroups x kTileRows float4 groups of U, first lists the parts whose windows touch the
// tile (in part order, so that the result does not depend on the tiling), and its threads
// then visit only those instead of all of them.
constexpr int kTileGroups = 16, kTileRows = 16;
constexpr int kTiledMergeParts = 16;      // frames of at least this many parts take the tiled merge
static_assert(kTileGroups * kTileRows == kMergeThreads && kMaxParts <= kMergeThreads, "one test per thread");

template <bool IS_MAX>
__global__ void __launch_bounds__(kMergeThreads)
k_window_merge_tiled(MergeArgs a) {
  __shared__ Win16 lwin[kMaxParts];
  __shared__ short lpart[kMaxParts];
  __shared__ int wave_hits[kMergeThreads / 64];
  const int fcl = blockIdx.y;
  const int bl = fcl / a.oc, b = a.b0 + bl;
  const int fc = fcl + a.b0 * a.oc;
  const size_t fo = (size_t)b * a.oc_total + a.ch0 + (fcl - bl * a.oc);
  const Window U = widen(a.unions[bl]);
  const int win_stride = a.win_stride;
  const int tiles_x = ((U.w >> 2) + kTileGroups - 1) / kTileGroups;
  const int tiles_z = (U.h + kTileRows - 1) / kTileRows;
  const int t = blockIdx.x;
  if (t >= tiles_x * tiles_z) {                  // the whole block
    signal_slot_free(a.signal, a.ticket);
    return;
  }
  const int tz = t / tiles_x, tx = t - tz * tiles_x;
  const int x0 = U.x0 + tx * (kTileGroups * 4), z0 = U.z0 + tz * kTileRows;
  const int x1 = min(x0 + kTileGroups * 4, U.x0 + U.w), z1 = min(z0 + kTileRows, U.z0 + U.h);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Win16 mine = Win16{0, 0, 0, 0};
  bool hit = false;
  if (tid < a.nparts) {
    mine = a.wins[bl * win_stride + tid];
    hit = mine.w > 0 && mine.x0 < x1 && mine.x0 + mine.w > x0 && mine.z0 < z1 && mine.z0 + mine.h > z0;
  }
  const unsigned long long votes = __builtin_amdgcn_ballot_w64(hit);
  if (lane == 0) wave_hits[wave] = __builtin_popcountll(votes);
  __syncthreads();
  int before = 0, n = 0;
#pragma unroll
  for (int w = 0; w < kMergeThreads / 64; ++w) {
    before += w < wave ? wave_hits[w] : 0;
    n += wave_hits[w];
  }
  if (hit) {
    const int at = before + __builtin_popcountll(votes & ((1ull << lane) - 1ull));
    lwin[at] = mine;
    lpart[at] = (short)tid;
  }
  __syncthreads();

  const int x = x0 + ((tid & (kTileGroups - 1)) << 2), zb = z0 + tid / kTileGroups;
  if (x >= x1 || zb >= z1) {
    signal_slot_free(a.signal, a.ticket);
    return;
  }
  float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
  const float* slabs = a.slabs + (size_t)fc * a.nparts * a.slab_stride;
  for (int j = 0; j < n; ++j) {
    const Window w = widen(lwin[j]);
    const unsigned ux = (unsigned)(x - w.x0), uz = (unsigned)(zb - w.z0);
    if (ux >= (unsigned)w.w || uz >= (unsigned)w.h) continue;
    const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)lpart[j] * a.slab_stride +
                                                     (size_t)uz * w.w + ux);
    acc.x = IS_MAX ? fmaxf(acc.x, v.x) : fminf(acc.x, v.x);
    acc.y = IS_MAX ? fmaxf(acc.y, v.y) : fminf(acc.y, v.y);
    acc.z = IS_MAX ? fmaxf(acc.z, v.z) : fminf(acc.z, v.z);
    acc.w = IS_MAX ? fmaxf(acc.w, v.w) : fminf(acc.w, v.w);
  }
  const size_t cell = fo * (size_t)a.mh * a.mw + (size_t)zb * a.mw + x;
  *reinterpret_cast<float4*>(a.out + cell) = acc;
  const uint32_t mk = (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
                      ((uint32_t)mask_of(acc.z, a.fill) << 16) |
                      ((uint32_t)mask_of(acc.w, a.fill) << 24);
  *reinterpret_cast<uint32_t*>(a.mask + cell) = mk;
  signal_slot_free(a.signal, a.ticket);
}

// Batch fuse (north_star "projected+fused"): fused[c] = max/min over the frames
// whose union window covers c of out[b][c]; frames that do not cover c hold the
// fill value there, so they cannot change the result and are never read.
struct FuseArgs {
  int B, dc, mh, mw;          // B frames in this launch, starting at frame b0
  int b0, accumulate;         // accumulate: fold into the current content of `fused`
  float fill;
  const Win16* unions;        // (B_total)   written by k_window_scatter
  const float* maps;          // (B_total, dc, mh, mw)
  float* fused;               // (dc, mh, mw)
  uint8_t* fused_mask;
};

constexpr int kFuseGroups = 32;     // float4 groups of the fused map per block
constexpr int kFuseLanes = 8;       // threads sharing one group, frames b = lane (mod 8)

// Block = 32 groups x 8 frame lanes.  Each thread tests the unions of its frames
// (lane, lane + 8, ...) eight at a time, loads the covered maps (independent
// 16-byte loads), and the 8 partial results of a group are combined through LDS.
template <bool IS_MAX>
__global__ void __launch_bounds__(kFuseGroups * kFuseLanes)
k_fuse_unions(FuseArgs a) {
  __shared__ float4 part[kFuseLanes][kFuseGroups];
  const int ch = blockIdx.y;
  const int g4 = a.mw >> 2;
  const int gi = threadIdx.x & (kFuseGroups - 1), lane = threadIdx.x / kFuseGroups;
  const int g = blockIdx.x * kFuseGroups + gi;
  const bool live = g < g4 * a.mh;
  const int z = live ? g / g4 : 0, x = live ? (g - z * g4) << 2 : 0;
  const size_t M = (size_t)a.mh * a.mw;
  const size_t cell = (size_t)z * a.mw + x;
  float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
  if (a.accumulate && lane == 0 && live)
    acc = *reinterpret_cast<const float4*>(a.fused + (size_t)ch * M + cell);
  for (int b0 = lane; b0 < a.B; b0 += 8 * kFuseLanes) {
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int bb = b0 + k * kFuseLanes;
      v[k] = acc;
      if (bb < a.B) {
        const Window U = widen(a.unions[a.b0 + bb]);
        if ((unsigned)(z - U.z0) < (unsigned)U.h && (unsigned)(x - U.x0) < (unsigned)U.w)
          v[k] = *reinterpret_cast<const float4*>(
              a.maps + ((size_t)(a.b0 + bb) * a.dc + ch) * M + cell);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      acc.x = IS_MAX ? fmaxf(acc.x, v[k].x) : fminf(acc.x, v[k].x);
      acc.y = IS_MAX ? fmaxf(acc.y, v[k].y) : fminf(acc.y, v[k].y);
      acc.z = IS_MAX ? fmaxf(acc.z, v[k].z) : fminf(acc.z, v[k].z);
      acc.w = IS_MAX ? fmaxf(acc.w, v[k].w) : fminf(acc.w, v[k].w);
    }
  }
  part[lane][gi] = acc;
  __syncthreads();
  if (lane == 0 && live) {
#pragma unroll
    for (int k = 1; k < kFuseLanes; ++k) {
      const float4 o = part[k][gi];
      acc.x = IS_MAX ? fmaxf(acc.x, o.x) : fminf(acc.x, o.x);
      acc.y = IS_MAX ? fmaxf(acc.y, o.y) : fminf(acc.y, o.y);
      acc.z = IS_MAX ? fmaxf(acc.z, o.z) : fminf(acc.z, o.z);
      acc.w = IS_MAX ? fmaxf(acc.w, o.w) : fminf(acc.w, o.w);
    }
    *reinterpret_cast<float4*>(a.fused + (size_t)ch * M + cell) = acc;
    const uint32_t mk = (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
                        ((uint32_t)mask_of(acc.z, a.fill) << 16) |
                        ((uint32_t)mask_of(acc.w, a.fill) << 24);
    *reinterpret_cast<uint32_t*>(a.fused_mask + (size_t)ch * M + cell) = mk;
  }
}

// Direct batch fuse from the slabs (dm_orth_project_fused_f32): fused[c] =
// max/min over every (frame, part) window covering c of its slab value -- the
// per-frame maps are never materialised.  Same block shape as k_fuse_unions; the
// 8 lanes of a group split the B * nparts windows.
struct FuseWinArgs {
  int nwin;                   // windows of this launch: (frames of the chunk) * nparts
  int b0;                     // first frame of the chunk
  int nparts, oc, ch0, oc_total, mh, mw;
  int slab_stride;
  int accumulate;
  int gx0, gz0, gx1, gz1;     // bounding box of every window of the call (cells, half open)
  float fill;
  const Win16* wins;          // (B_total, nparts)   written by k_window_scatter
  uint32_t* signal;           // table ring (see MergeArgs)
  uint32_t ticket;
  const float* slabs;         // ((b * oc + chl) * nparts + p) * slab_stride
  float* fused;               // (oc_total, mh, mw)
  uint8_t* fused_mask;
};

constexpr int kFuseChunk = 1024;    // windows examined per candidate-list round

template <bool IS_MAX>
__global__ void __launch_bounds__(kFuseGroups * kFuseLanes)
k_fuse_windows(FuseWinArgs a) {
  __shared__ float4 part[kFuseLanes][kFuseGroups];
  __shared__ int4 cwin[kFuseChunk];     // candidate windows {x0, z0, w, h} ...
  __shared__ int cslab[kFuseChunk];     // ... and their slab index
  __shared__ int ncand;
  const int chl = blockIdx.y, ch = a.ch0 + chl;
  const int g4 = a.mw >> 2;
  const int total = g4 * a.mh;
  const size_t M = (size_t)a.mh * a.mw;
  // Most of a large global map is out of reach of the whole call.  The first `heavy`
  // blocks own the 32-group tiles of the call's bounding box (one map row each); the
  // others stream the fill value (or, accumulating, only refresh the mask) over
  // everything outside it, 8 groups per thread: few, fat blocks instead of one tiny
  // block per 128 cells of a mostly empty map.
  const int bw4 = (a.gx1 - a.gx0) >> 2;                           // groups per bounding-box row
  const int tpr = (bw4 + kFuseGroups - 1) / kFuseGroups;          // tiles per bounding-box row
  const int heavy = a.gx1 > a.gx0 ? tpr * (a.gz1 - a.gz0) : 0;
  if ((int)blockIdx.x >= heavy) {                                 // block-uniform
    const int first = ((int)blockIdx.x - heavy) * (int)blockDim.x * 8 + (int)threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int g = first + k * (int)blockDim.x;
      if (g >= total) break;
      const int z = g / g4, x = (g - z * g4) << 2;
      if (z >= a.gz0 && z < a.gz1 && x >= a.gx0 && x < a.gx1) continue;     // a tile's cell
      const size_t cell = (size_t)ch * M + (size_t)z * a.mw + x;
      uint32_t mk = 0u;
      if (a.accumulate) {
        const float4 v = *reinterpret_cast<const float4*>(a.fused + cell);
        mk = (uint32_t)mask_of(v.x, a.fill) | ((uint32_t)mask_of(v.y, a.fill) << 8) |
             ((uint32_t)mask_of(v.z, a.fill) << 16) | ((uint32_t)mask_of(v.w, a.fill) << 24);
      } else {
        *reinterpret_cast<float4*>(a.fused + cell) = make_float4(a.fill, a.fill, a.fill, a.fill);
      }
      *reinterpret_cast<uint32_t*>(a.fused_mask + cell) = mk;
    }
    signal_slot_free(a.signal, a.ticket);
    return;
  }
  const int gi = threadIdx.x & (kFuseGroups - 1), lane = threadIdx.x / kFuseGroups;
  const int trow = (int)blockIdx.x / tpr, tcol = (int)blockIdx.x - trow * tpr;
  const int z = a.gz0 + trow;
  const int gx = (a.gx0 >> 2) + tcol * kFuseGroups + gi;          // group index within the row
  const bool live = gx < (a.gx1 >> 2);
  const int x = (live ? gx : (a.gx0 >> 2)) << 2;
  const int z_lo = z, z_hi = z;
  const int x_lo = ((a.gx0 >> 2) + tcol * kFuseGroups) << 2;
  const int x_hi = x_lo + 4 * kFuseGroups < a.gx1 ? x_lo + 4 * kFuseGroups : a.gx1;
  const size_t cell = (size_t)ch * M + (size_t)z * a.mw + x;
  float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
  if (a.accumulate && lane == 0 && live) acc = *reinterpret_cast<const float4*>(a.fused + cell);
  for (int c0 = 0; c0 < a.nwin; c0 += kFuseChunk) {
    // round 1: which windows of this chunk touch the block's cells at all?
    if (threadIdx.x == 0) ncand = 0;
    __syncthreads();
    for (int r = c0 + threadIdx.x; r < a.nwin && r < c0 + kFuseChunk; r += blockDim.x) {
      const Window w = widen(a.wins[(size_t)a.b0 * a.nparts + r]);
      if (w.w > 0 && w.z0 <= z_hi && w.z0 + w.h > z_lo && w.x0 < x_hi && w.x0 + w.w > x_lo) {
        const int slot = atomicAdd(&ncand, 1);
        const int b = r / a.nparts, p = r - b * a.nparts;
        cwin[slot] = make_int4(w.x0, w.z0, w.w, w.h);
        cslab[slot] = ((a.b0 + b) * a.oc + chl) * a.nparts + p;
      }
    }
    __syncthreads();
    // round 2: the 8 lanes of a group split the candidates, four slab loads in flight
    const int n = ncand;
    for (int i0 = lane; i0 < n; i0 += 4 * kFuseLanes) {
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * kFuseLanes;
        v[k] = acc;
        if (i < n) {
          const int4 w = cwin[i];
          const unsigned ux = (unsigned)(x - w.x), uz = (unsigned)(z - w.y);
          if (ux < (unsigned)w.z && uz < (unsigned)w.w)
            v[k] = *reinterpret_cast<const float4*>(
                a.slabs + (size_t)cslab[i] * a.slab_stride + (size_t)uz * w.z + ux);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc.x = IS_MAX ? fmaxf(acc.x, v[k].x) : fminf(acc.x, v[k].x);
        acc.y = IS_MAX ? fmaxf(acc.y, v[k].y) : fminf(acc.y, v[k].y);
        acc.z = IS_MAX ? fmaxf(acc.z, v[k].z) : fminf(acc.z, v[k].z);
        acc.w = IS_MAX ? fmaxf(acc.w, v[k].w) : fminf(acc.w, v[k].w);
      }
    }
    __syncthreads();
  }
  part[lane][gi] = acc;
  __syncthreads();
  if (lane == 0 && live) {
#pragma unroll
    for (int k = 1; k < kFuseLanes; ++k) {
      const float4 o = part[k][gi];
      acc.x = IS_MAX ? fmaxf(acc.x, o.x) : fminf(acc.x, o.x);
      acc.y = IS_MAX ? fmaxf(acc.y, o.y) : fminf(acc.y, o.y);
      acc.z = IS_MAX ? fmaxf(acc.z, o.z) : fminf(acc.z, o.z);
      acc.w = IS_MAX ? fmaxf(acc.w, o.w) : fminf(acc.w, o.w);
    }
    *reinterpret_cast<float4*>(a.fused + cell) = acc;
    const uint32_t mk = (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
                        ((uint32_t)mask_of(acc.z, a.fill) << 16) |
                        ((uint32_t)mask_of(acc.w, a.fill) << 24);
    *reinterpret_cast<uint32_t*>(a.fused_mask + cell) = mk;
  }
  signal_slot_free(a.signal, a.ticket);
}

}  // namespace

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

bool window_path_supported(const dm_params& p) {
  if (p.reduction != DM_REDUCE_MAX && p.reduction != DM_REDUCE_MIN) return false;
  if (p.mw % 4 != 0) return false;
  if (p.mw > 32767 || p.mh > 32767) return false;   // Win16
  if (!(p.fill == p.fill)) return false;       // NaN fill has no order
  return true;
}

// Device copies of the call's geometry, written by k_window_scatter for the kernels that
// follow it: part windows, union windows.
static int frames_per_chunk(int nparts) {
  const int win_stride = nparts <= kFewParts ? kFewParts : nparts;
  const int chunk = kChunkWins / win_stride;
  return chunk > kChunkFrames ? kChunkFrames : chunk;
}

static size_t geometry_bytes(int B, int nparts) {
  const int chunk = frames_per_chunk(nparts);
  return align_up((size_t)B * nparts * sizeof(Win16), 256) + align_up((size_t)B * sizeof(Win16), 256) +
         align_up((size_t)((B + chunk - 1) / chunk) * sizeof(ScatterTables), 256);
}

static constexpr size_t kSlabBudget = (size_t)256 << 20;   // slab bytes per channel group

size_t window_workspace_bytes(const dm_params& p) {
  // slabs of one channel group (+ the scratch mask of the height pass).  Sized for up
  // to 4x the default number of parts (run_window splits further only when windows do
  // not fit in LDS, and falls back to the generic path if the workspace cannot hold that).
  size_t cap = (size_t)p.mh * p.mw;
  if (cap > kMaxLdsBytes / 4) cap = kMaxLdsBytes / 4;
  const Parts d = choose_parts(p, 1);
  size_t np = (size_t)d.pc * d.pr * 4;
  if (np > kMaxParts) np = kMaxParts;
  const size_t oc = p.vc ? p.vc : p.dc;
  const size_t one = (size_t)p.B * np * align_up(cap, 4) * 4;     // one channel of every frame
  size_t slabs = one * oc;
  if (slabs > kSlabBudget) slabs = one > kSlabBudget ? one : kSlabBudget;
  const size_t height_mask = p.vc ? align_up((size_t)p.B * p.dc * p.mh * p.mw, 256) : 0;
  return geometry_bytes(p.B, kMaxParts) + slabs + height_mask;
}

namespace {

struct Staged {                 // what run_window keeps between its passes
  Parts parts;
  int nparts, slab_stride, max_union;
  int max_tiles;                // k_window_merge_tiled blocks of the largest union window
  int gx0, gz0, gx1, gz1;       // bounding box of all union windows (empty: gx1 <= gx0)
  Win16* g_wins;                // device copies (workspace head)
  Win16* g_unions;
  const ScatterTables* d_tables;  // one per chunk of frames
  // d_tables is a slot of the thread's table ring (device memory the host wrote through the
  // PCIe BAR, no copy operation): only k_window_scatter may read it, the kernel that follows
  // signals the slot free.  Else: the workspace, filled by a stream-ordered copy.
  bool ring_tables;
  uint32_t* slot_done;            // pinned host word the signalling kernel stores its ticket to
  uint32_t* slot_issued;          // host: ticket of the last signalling launch that used the slot
  uint32_t* ring_ticket;          // host: the ring's ticket counter
  int chunk;                    // frames per chunk
  size_t geom_bytes;
  bool fast, fast_div;
  float res_inv, fx_inv, fy_inv;
  const FrameRec* frames;       // (B)            host, thread-local storage
  const Win16* wins;            // (B, nparts)
  const Win16* unions;          // (B)
};

template <class K, class... Args>
inline hipError_t launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s,
                         const Args&... args) {
  hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
  return hipGetLastError();
}

using Kernel = void (*)(ScatterArgs, const ScatterTables*);

Kernel pick_kernel(bool is_max, bool fast, bool has_valid, bool has_value, bool vec4, bool lean) {
#define DM_K(M, F, V, S) {k_window_scatter<M, F, V, S, 1, false>, k_window_scatter<M, F, V, S, 4, false>}
  // [is_max][fast][has_valid][has_value][vec4]
  static const Kernel table[2][2][2][2][2] = {
      {{{DM_K(false, false, false, false), DM_K(false, false, false, true)},
        {DM_K(false, false, true, false), DM_K(false, false, true, true)}},
       {{DM_K(false, true, false, false), DM_K(false, true, false, true)},
        {DM_K(false, true, true, false), DM_K(false, true, true, true)}}},
      {{{DM_K(true, false, false, false), DM_K(true, false, false, true)},
        {DM_K(true, false, true, false), DM_K(true, false, true, true)}},
       {{DM_K(true, true, false, false), DM_K(true, true, false, true)},
        {DM_K(true, true, true, false), DM_K(true, true, true, true)}}}};
#undef DM_K
  static const Kernel lean_table[2][2] = {
      {k_window_scatter<false, true, false, false, 4, true>,
       k_window_scatter<false, true, false, true, 4, true>},
      {k_window_scatter<true, true, false, false, 4, true>,
       k_window_scatter<true, true, false, true, 4, true>}};
  return lean ? lean_table[is_max][has_value] : table[is_max][fast][has_valid][has_value][vec4];
}

// One pass: scatter `value` (or the heights when NULL) of channels [0, oc_total)
// into out/mask, channel group by channel group, chunk of frames by chunk of frames.
// With `fused` set, the per-frame maps are skipped (out/mask NULL) and every channel
// group's slabs are reduced straight into the (oc_total, mh, mw) fused map.
hipError_t window_pass(const dm_params& p, const Staged& st, float* slabs,
                       const float* depth, const float* value, const uint8_t* valid, float* out,
                       uint8_t* mask, int oc_total, float fill, bool is_max, size_t slab_bytes,
                       hipStream_t s, float* fused = nullptr, uint8_t* fused_mask = nullptr,
                       int accumulate = 0) {
  ScatterArgs sa;
  sa.W = p.W; sa.H = p.H;
  sa.clip = p.clip_border > 0 ? p.clip_border : 0;
  sa.flip_h = p.flip_h != 0;
  sa.cx = p.cx; sa.cy = p.cy; sa.fx = p.fx; sa.fy = p.fy; sa.res = p.res;
  sa.res_inv = st.res_inv; sa.fx_inv = st.fx_inv; sa.fy_inv = st.fy_inv;
  sa.dmin = p.has_dmin ? p.dmin : -INFINITY;
  sa.dmax = p.has_dmax ? p.dmax : INFINITY;
  sa.hmax = p.has_hmax ? p.hmax : INFINITY;
  sa.Hm1 = (float)(p.H - 1); sa.mhm1 = (float)(p.mh - 1);
  sa.parts = st.parts;
  sa.dc = p.dc; sa.valid_c = p.valid_c;
  sa.oc_total = oc_total;
  sa.slab_stride = st.slab_stride;
  sa.fill = fill;
  sa.depth = depth; sa.value = value; sa.valid = valid;
  sa.slabs = slabs;
  sa.out = out; sa.mask = mask; sa.mh = p.mh; sa.mw = p.mw;

  const bool has_valid = valid != nullptr, has_value = value != nullptr;
  const bool vec4 = (p.W % 4 == 0) && (reinterpret_cast<uintptr_t>(depth) % 16 == 0) &&
                    (!value || reinterpret_cast<uintptr_t>(value) % 16 == 0) &&
                    (st.parts.wp % 4 == 0);
  sa.g_wins = st.g_wins; sa.g_unions = st.g_unions;
#ifdef DM_STAMPS
  sa.stamps = g_stamp_buffer;
#endif
  const size_t lds_bytes = align_up((size_t)st.slab_stride * 4, 16) + 64 * 4;   // + dummy cells
  // lean variant: both depth bounds finite, no height truncation, no border, no valid map
  const bool lean = st.fast && vec4 && !has_valid && p.has_dmin && p.has_dmax &&
                    isfinite(p.dmin) && isfinite(p.dmax) && !p.has_hmax && p.clip_border <= 0;
  const Kernel kfn = pick_kernel(is_max, st.fast, has_valid, has_value, vec4, lean);
  hipError_t e = hipSuccess;
  {   // raise the dynamic-LDS limit once per kernel variant and device
    static thread_local const void* done[64][8] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void* key = reinterpret_cast<const void*>(kfn);
    bool seen = false;
    int free_slot = -1;
    if (dev >= 0 && dev < 8) {
      for (int i = 0; i < 64; ++i) {
        if (done[i][dev] == key) { seen = true; break; }
        if (!done[i][dev] && free_slot < 0) free_slot = i;
      }
    }
    if (!seen) {
      e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 8 && free_slot >= 0) done[free_slot][dev] = key;
    }
  }

  const int chunk = st.chunk;                // frames per launch: what one table holds
  // channel groups: the slabs of one group fit the workspace's slab region
  const size_t per_channel = (size_t)p.B * st.nparts * st.slab_stride * 4;
  int group = (int)(slab_bytes / (per_channel ? per_channel : 1));
  if (group < 1) group = 1;
  if (group > oc_total) group = oc_total;
  if (group > 65535 / chunk) group = 65535 / chunk;       // merge grid.y = frames * channels
  for (int ch0 = 0; ch0 < oc_total; ch0 += group) {
    const int oc = oc_total - ch0 < group ? oc_total - ch0 : group;
    sa.oc = oc; sa.ch0 = ch0;
    for (int b0 = 0; b0 < p.B; b0 += chunk) {
      const int nb = p.B - b0 < chunk ? p.B - b0 : chunk;
      sa.b0 = b0;
      e = launch(kfn, dim3(st.nparts, oc, nb), dim3(kScatterThreads), lds_bytes, s, sa,
                 st.d_tables + b0 / chunk);
      if (e != hipSuccess) return e;
      if (!fused && st.max_union > 0) {
        MergeArgs ma;
        ma.b0 = b0; ma.oc = oc; ma.ch0 = ch0; ma.oc_total = oc_total; ma.mh = p.mh; ma.mw = p.mw;
        ma.nparts = st.nparts; ma.slab_stride = st.slab_stride; ma.fill = fill;
        ma.signal = nullptr; ma.ticket = 0;
        if (st.ring_tables) {
          ma.wins = st.g_wins + (size_t)b0 * st.nparts; ma.unions = st.g_unions + b0;
          ma.win_stride = st.nparts;
          ma.signal = st.slot_done;
          ma.ticket = *st.slot_issued = ++*st.ring_ticket;
        } else {
          const ScatterTables* t = st.d_tables + b0 / chunk;     // device address arithmetic only
          ma.wins = t->wins; ma.unions = t->unions;
          ma.win_stride = st.nparts <= kFewParts ? kFewParts : st.nparts;
        }
        ma.slabs = slabs; ma.out = out; ma.mask = mask;
        if (st.nparts >= kTiledMergeParts) {
          const dim3 g((unsigned)st.max_tiles, nb * oc);
          e = is_max ? launch(k_window_merge_tiled<true>, g, dim3(kMergeThreads), 0, s, ma)
                     : launch(k_window_merge_tiled<false>, g, dim3(kMergeThreads), 0, s, ma);
        } else {
          const dim3 g((unsigned)((st.max_union / 4 + kMergeThreads - 1) / kMergeThreads), nb * oc);
          e = is_max ? launch(k_window_merge<true>, g, dim3(kMergeThreads), 0, s, ma)
                     : launch(k_window_merge<false>, g, dim3(kMergeThreads), 0, s, ma);
        }
        if (e != hipSuccess) return e;
      }
    }
    if (fused) {     // every frame's slabs of this channel group are in place
      FuseWinArgs fa;
      fa.nwin = p.B * st.nparts; fa.b0 = 0; fa.nparts = st.nparts; fa.oc = oc; fa.ch0 = ch0;
      fa.oc_total = oc_total; fa.mh = p.mh; fa.mw = p.mw; fa.slab_stride = st.slab_stride;
      fa.accumulate = accumulate; fa.fill = fill;
      fa.gx0 = st.gx0; fa.gz0 = st.gz0; fa.gx1 = st.gx1; fa.gz1 = st.gz1;
      fa.wins = st.g_wins; fa.slabs = slabs; fa.fused = fused; fa.fused_mask = fused_mask;
      fa.signal = nullptr; fa.ticket = 0;
      if (st.ring_tables) {
        fa.signal = st.slot_done;
        fa.ticket = *st.slot_issued = ++*st.ring_ticket;
      }
      const int bw4 = (st.gx1 - st.gx0) / 4;
      const int heavy = st.gx1 > st.gx0 ? ((bw4 + kFuseGroups - 1) / kFuseGroups) * (st.gz1 - st.gz0) : 0;
      const int per_fill_block = kFuseGroups * kFuseLanes * 8;
      const int fill_blocks = (int)(((size_t)p.mh * p.mw / 4 + per_fill_block - 1) / per_fill_block);
      const dim3 g((unsigned)(heavy + fill_blocks), oc);
      const dim3 blk(kFuseGroups * kFuseLanes);
      e = is_max ? launch(k_fuse_windows<true>, g, blk, 0, s, fa)
                 : launch(k_fuse_windows<false>, g, blk, 0, s, fa);
      if (e != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

// What the previous calls of a shape settled on: the split into parts (stage_windows) and
// whether the frames have to go through in halves (run_window).  Per thread, a few shapes.
struct Remembered {
  dm_params key;
  Parts parts;
  bool valid, halve;
  int hopeless;          // calls left before the search is tried again after it failed
};
Remembered& remembered(const dm_params& p) {
  thread_local Remembered slots[4] = {};
  thread_local int next = 0;
  for (Remembered& r : slots)
    if (memcmp(&r.key, &p, sizeof(dm_params)) == 0) return r;
  Remembered& r = slots[next];
  next = (next + 1) % 4;
  r = Remembered{};
  r.key = p;
  return r;
}

thread_local int g_last_split[4] = {0, 0, 0, 0};   // dm_debug_last_split
thread_local bool g_force_bands = false;           // dm_debug_force_bands

// Depth bands multiply the workgroups, each with a window to initialise and flush: with few
// pixels per part that costs more than the generic path's atomics.  Both sides as measured on
// MI355X: generic = 22 ps per point + fill at 3 TB/s; windowed = per pass
// waves of 256 workgroups at 4 us + (slab + part pixels) at 20 GB/s per CU, plus the merge
// reading the slabs at 3.7 TB/s.
bool banded_split_pays(const dm_params& p, const Parts& parts, int nparts, int max_area,
                       size_t slab_capacity) {
  const double oc = p.vc ? p.vc : p.dc;
  const double points = (double)p.B * p.H * p.W * oc, cells = (double)p.B * oc * p.mh * p.mw;
  const double t_generic = points * 22e-6 + cells * 5.0 / 3.0e6;
  const double slab = (double)align_up((size_t)max_area, 4) * 4.0;
  double frames = p.B;
  int passes = 1;
  while (frames > 1.0 && frames * nparts * slab > (double)slab_capacity) {
    frames = ceil(frames / 2.0);
    passes *= 2;
  }
  const double t_wg = 4.0 + (slab + (double)parts.wp * parts.hp * 4.0) / 20480.0;
  const double waves = ceil(frames * oc * nparts / 256.0);
  const double t_window = passes * (waves * t_wg + 6.0) + (double)p.B * oc * nparts * slab / 3.7e6;
  return t_window < t_generic;
}

// The scatter tables of a call are ~10 KB the kernels need before they can start.  Staging
// them with a stream-ordered copy costs a copy kernel and a dependent launch (~4 us of every
// call).  On a large-BAR system the host instead writes them straight into device memory: a
// per-thread, per-device ring of kTableSlots fine-grained device buffers (1.4 MB, allocated on
// first use, kept for the life of the process -- the library's only persistent allocation).
// PCIe keeps posted writes ordered ahead of the packet fetch that starts the kernel.  A slot
// is reused only after the kernel that follows its k_window_scatter in stream order has
// stored the slot's ticket to pinned host memory; the host waits for that (bounded), and
// without a free slot or a large BAR the call takes the staged copy.
constexpr int kTableSlots = 64;
struct TableRing {
  ScatterTables* slots = nullptr;            // device memory, host-writable
  volatile uint32_t* done = nullptr;         // pinned host memory, written by the GPU
  uint32_t issued[kTableSlots] = {};         // ticket of the last signalling launch per slot
  uint32_t ticket = 0;
  int next = 0;
  bool tried = false, off = false;

  int acquire() {
    if (!tried) {
      tried = true;
      static const bool disabled = getenv("DM_NO_TABLE_RING") != nullptr;
      int dev = 0, large_bar = 0;
      void* d = nullptr;
      void* h = nullptr;
      if (disabled || hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev) != hipSuccess ||
          !large_bar ||
          hipExtMallocWithFlags(&d, kTableSlots * sizeof(ScatterTables), hipDeviceMallocFinegrained) !=
              hipSuccess) {
        (void)hipGetLastError();
        off = true;
      } else if (hipHostMalloc(&h, kTableSlots * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(d);
        off = true;
      } else {
        slots = static_cast<ScatterTables*>(d);
        done = static_cast<volatile uint32_t*>(h);
        for (int i = 0; i < kTableSlots; ++i) done[i] = 0;
      }
    }
    if (off) return -1;
    const int i = next;
    if (done[i] != issued[i]) {              // the GPU is kTableSlots calls behind: wait for it
      const auto t0 = std::chrono::steady_clock::now();
      while (done[i] != issued[i]) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
          // calls of several ms each (64 of them in flight), or a launch that failed after
          // its slot was taken: this thread goes back to the staged copy for good -- 4 us
          // per call are nothing to such calls, and nobody waits here twice
          off = true;
          return -1;
        }
        __builtin_ia32_pause();
      }
    }
    next = (next + 1) % kTableSlots;
    return i;
  }
};
TableRing& table_ring() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  thread_local TableRing rings[16];
  return rings[dev >= 0 && dev < 16 ? dev : 0];
}

// Host-side geometry of a call: parts, part windows, frame records.  Returns
// hipErrorNotSupported when the windows cannot be made to fit in LDS (the caller then
// takes the generic path); nothing has been enqueued in that case.
hipError_t stage_windows(const dm_params& p, const dm_frame* frames_host, void* ws,
                         size_t ws_bytes, Staged& st, size_t& slab_bytes, hipStream_t s,
                         hipEvent_t before = nullptr, bool will_fuse_windows = false) {
  if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return hipErrorNotSupported;
  thread_local std::vector<FrameRec> recs;
  thread_local std::vector<Win16> wins;      // (B, nparts) then (B) unions
  // more, narrower parts until every window fits in LDS; when splitting the image does not
  // get there (a long thin wedge: fine resolution, long range) the depth range is split too
  int max_area = 0;
  const bool can_band = p.has_dmin && p.has_dmax && p.dmin >= 0.0f && p.dmax > p.dmin &&
                        isfinite(p.dmax);
  thread_local std::vector<PartSlopes> slopes;
  const bool bounded = frustum_bounded(p);
  auto evaluate = [&](const Parts& parts) -> bool {
    const int pd = parts.pd;
    st.parts = parts;
    const int image_parts = parts.pc * parts.pr;
    st.nparts = image_parts * pd;
    wins.resize((size_t)p.B * (st.nparts + 1));
    Win16* unions = wins.data() + (size_t)p.B * st.nparts;
    slopes.resize(image_parts);
    for (int pr = 0; pr < parts.pr; ++pr)
      for (int pc = 0; pc < parts.pc; ++pc) {
        const int q0 = pc * parts.wp, r0 = pr * parts.hp;
        const int q1 = q0 + parts.wp < p.W ? q0 + parts.wp : p.W;
        const int r1 = r0 + parts.hp < p.H ? r0 + parts.hp : p.H;
        slopes[pr * parts.pc + pc] = part_slopes(p, q0, q1, r0, r1);
      }
    max_area = 0; st.max_union = 0; st.max_tiles = 0;
    st.gx0 = p.mw; st.gz0 = p.mh; st.gx1 = 0; st.gz1 = 0;
    for (int b = 0; b < p.B; ++b) {
      int ux0 = p.mw, ux1 = 0, uz0 = p.mh, uz1 = 0;
      const FrameAffine fa = frame_affine(p, frames_host[b]);
      Win16* row = wins.data() + (size_t)b * st.nparts;
      for (int k = 0; k < pd; ++k) {
        float dlo = p.dmin, dhi = p.dmax;
        if (pd > 1) band_bounds(p.dmin, p.dmax, pd, k, dlo, dhi);
        for (int ip = 0; ip < image_parts; ++ip) {
          const Window w = part_window(p, fa, slopes[ip], bounded, dlo, dhi);
          row[k * image_parts + ip] = narrow(w);
          if (w.w * w.h > max_area) max_area = w.w * w.h;
          if (w.w > 0) {
            if (w.x0 < ux0) ux0 = w.x0;
            if (w.x0 + w.w > ux1) ux1 = w.x0 + w.w;
            if (w.z0 < uz0) uz0 = w.z0;
            if (w.z0 + w.h > uz1) uz1 = w.z0 + w.h;
          }
        }
      }
      const Window U = ux1 > ux0 ? Window{ux0, uz0, ux1 - ux0, uz1 - uz0} : Window{0, 0, 0, 0};
      unions[b] = narrow(U);
      if (U.w * U.h > st.max_union) st.max_union = U.w * U.h;
      const int tiles = ((U.w / 4 + kTileGroups - 1) / kTileGroups) * ((U.h + kTileRows - 1) / kTileRows);
      if (tiles > st.max_tiles) st.max_tiles = tiles;
      if (U.w > 0) {
        if (U.x0 < st.gx0) st.gx0 = U.x0;
        if (U.z0 < st.gz0) st.gz0 = U.z0;
        if (U.x0 + U.w > st.gx1) st.gx1 = U.x0 + U.w;
        if (U.z0 + U.h > st.gz1) st.gz1 = U.z0 + U.h;
      }
    }
    static const bool verbose = getenv("DM_DEBUG_WINDOWS") != nullptr;
    if (verbose)
      fprintf(stderr, "[dm] pd=%d parts=%dx%d nparts=%d max_area=%d max_union=%d\n", pd, parts.pc,
              parts.pr, st.nparts, max_area, st.max_union);
    return (size_t)max_area * 4 + 64 * 4 + 16 <= (size_t)kMaxLdsBytes;
  };
  // the split that worked for the previous call of the same shape is tried first (the answer
  // moves with the poses only): one evaluation per call in the steady state
  Remembered& last = remembered(p);
  if (last.hopeless > 0 && !g_force_bands) {
    --last.hopeless;
    g_last_split[0] = g_last_split[1] = g_last_split[2] = 0;
    return hipErrorNotSupported;
  }
  const bool same_shape = last.valid && (last.parts.pd == 1 || can_band);
  bool fits = same_shape && evaluate(last.parts);
  if (!fits) {
    // candidates: the image splits without bands first (fewest parts first), then the banded
    // ones by their number of parts
    std::vector<Parts> cands;
    auto seen = [&](const Parts& c) {
      for (const Parts& o : cands)
        if (o.pc == c.pc && o.pr == c.pr && o.pd == c.pd) return true;
      return false;
    };
    for (int pd = 1; pd <= 8; pd *= 2) {
      if (pd > 1 && !can_band) break;
      for (int min_parts = 1;; min_parts *= 2) {
        const Parts c = choose_parts(p, min_parts, pd);
        const int image_parts = c.pc * c.pr;
        if (image_parts * pd > kMaxParts || image_parts < min_parts) break;   // cannot split further
        if (!seen(c)) cands.push_back(c);
        // a window that is the whole map cannot shrink by splitting the image
        if (!p.has_dmin || !p.has_dmax || image_parts >= 64) break;
      }
    }
    std::stable_sort(cands.begin(), cands.end(), [](const Parts& x, const Parts& y) {
      if ((x.pd == 1) != (y.pd == 1)) return x.pd == 1;
      return x.pc * x.pr * x.pd < y.pc * y.pr * y.pd;
    });
    for (const Parts& c : cands) {
      if (same_shape && c.pc == last.parts.pc && c.pr == last.parts.pr && c.pd == last.parts.pd) continue;
      if (evaluate(c)) {
        const size_t geom = geometry_bytes(p.B, st.nparts);
        fits = c.pd == 1 || g_force_bands || banded_split_pays(p, c, st.nparts, max_area, ws_bytes > geom ? ws_bytes - geom : 0);
        break;            // splits of more parts cost more still
      }
    }
  }
  last.valid = fits;
  if (fits) last.parts = st.parts;
  else last.hopeless = 63;       // the search is host time: not on every call of such a shape
  if (!fits) {
    g_last_split[0] = g_last_split[1] = g_last_split[2] = 0;
    return hipErrorNotSupported;
  }
  g_last_split[0] = st.parts.pc; g_last_split[1] = st.parts.pr; g_last_split[2] = st.parts.pd;
  st.slab_stride = (int)align_up((size_t)(max_area > 0 ? max_area : 4), 4);
  st.geom_bytes = geometry_bytes(p.B, st.nparts);
  if (ws_bytes < st.geom_bytes + (size_t)p.B * st.nparts * st.slab_stride * 4)
    return hipErrorOutOfMemory;       // run_window then takes the frames in two halves
  slab_bytes = ws_bytes - st.geom_bytes;
  {
    unsigned char* base = static_cast<unsigned char*>(ws);
    st.g_wins = reinterpret_cast<Win16*>(base);
    base += align_up((size_t)p.B * st.nparts * sizeof(Win16), 256);
    st.g_unions = reinterpret_cast<Win16*>(base);
    base += align_up((size_t)p.B * sizeof(Win16), 256);
    st.d_tables = reinterpret_cast<const ScatterTables*>(base);
  }

  recs.resize(p.B);
  for (int b = 0; b < p.B; ++b) {
    const dm_frame& f = frames_host[b];
    FrameRec& r = recs[b];
    memcpy(r.p, f.Rp, sizeof(r.p));
    r.cam_h = f.cam_height;
    r.wo = f.width_offset; r.ho = f.height_offset; r.pad = 0.0f;
    if (p.to_global) {
      memcpy(r.y, f.Ry, sizeof(r.y));
      r.tx = f.tx; r.tz = f.tz;
    } else {              // local map: neutral yaw, no translation (exact: x*1 + z*0 + 0)
      static const float eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      memcpy(r.y, eye, sizeof(eye));
      r.tx = 0.0f; r.tz = 0.0f;
    }
  }
  // branch-free exact division needs exactly rounded reciprocals and sane magnitudes
  st.fast_div = exact_reciprocal(p.res, &st.res_inv) && exact_reciprocal(p.fx, &st.fx_inv) &&
                exact_reciprocal(p.fy, &st.fy_inv) && p.res >= 1e-6f && p.res <= 1e6f &&
                p.fx >= 1e-6f && p.fx <= 1e6f && p.fy >= 1e-6f && p.fy <= 1e6f;
  if (!st.fast_div) st.res_inv = st.fx_inv = st.fy_inv = 0.0f;
  st.fast = st.fast_div && axis_aligned(recs.data(), p.B);
  st.frames = recs.data();
  st.wins = wins.data();
  st.unions = wins.data() + (size_t)p.B * st.nparts;
  // the launches' tables (thread-local staging: hipMemcpyAsync from pageable memory has
  // copied the bytes out by the time it returns)
  thread_local std::vector<ScatterTables> tabs;
  st.chunk = frames_per_chunk(st.nparts);
  const int win_stride = st.nparts <= kFewParts ? kFewParts : st.nparts;
  const int nchunks = (p.B + st.chunk - 1) / st.chunk;
  tabs.resize(nchunks);
  for (int c = 0; c < nchunks; ++c) {
    ScatterTables& tab = tabs[c];
    const int b0 = c * st.chunk;
    const int nb = p.B - b0 < st.chunk ? p.B - b0 : st.chunk;
    memcpy(tab.frames, st.frames + b0, (size_t)nb * sizeof(FrameRec));
    for (int i = 0; i < nb; ++i)
      memcpy(tab.wins + (size_t)i * win_stride, st.wins + (size_t)(b0 + i) * st.nparts,
             (size_t)st.nparts * sizeof(Win16));
    memcpy(tab.unions, st.unions + b0, (size_t)nb * sizeof(Win16));
  }
  if (before) {
    const hipError_t e = hipEventRecord(before, s);
    if (e != hipSuccess) return e;
  }
  const size_t table_bytes =
      nchunks > 1 ? (size_t)nchunks * sizeof(ScatterTables)
                  : offsetof(ScatterTables, wins) + (size_t)p.B * win_stride * sizeof(Win16);
  st.ring_tables = false;
  // a follow-up kernel has to exist to signal the slot free (no union window: no merge)
  if (nchunks == 1 && (will_fuse_windows || st.max_union > 0)) {
    TableRing& ring = table_ring();
    const int i = ring.acquire();
    if (i >= 0) {
      memcpy(ring.slots + i, tabs.data(), table_bytes);       // write-combined, through the BAR
      __builtin_ia32_sfence();                                // on the bus before the launch is
      st.d_tables = ring.slots + i;
      st.ring_tables = true;
      st.slot_done = const_cast<uint32_t*>(ring.done + i);
      st.slot_issued = ring.issued + i;
      st.ring_ticket = &ring.ticket;
      return hipSuccess;
    }
  }
  return hipMemcpyAsync(const_cast<ScatterTables*>(st.d_tables), tabs.data(), table_bytes,
                        hipMemcpyHostToDevice, s);
}

}  // namespace

hipError_t run_window(const dm_params& p, const dm_frame* frames_host, const float* depth,
                      const float* value, const uint8_t* valid, float* out, uint8_t* mask,
                      float* height, float* fused, uint8_t* fused_mask, void* ws,
                      size_t ws_bytes, hipEvent_t before_projection, hipEvent_t after_projection,
                      hipStream_t s) {
  const int oc_total = p.vc ? p.vc : p.dc;
  if (reinterpret_cast<uintptr_t>(out) % 16 != 0 || reinterpret_cast<uintptr_t>(mask) % 4 != 0 ||
      reinterpret_cast<uintptr_t>(fused) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(fused_mask) % 4 != 0 ||
      reinterpret_cast<uintptr_t>(height) % 16 != 0)
    return hipErrorNotSupported;
  Staged st;
  size_t slab_bytes = 0;
#ifdef DM_HOST_TIMING
  using clk = std::chrono::steady_clock;
  thread_local double acc_t[4] = {0, 0, 0, 0};
  thread_local int calls_t = 0;
  const auto t_0 = clk::now();
#endif
  Remembered& shape = remembered(p);
  if (shape.hopeless > 0 && !g_force_bands) {
    --shape.hopeless;
    g_last_split[0] = g_last_split[1] = g_last_split[2] = 0;
    return hipErrorNotSupported;
  }
  hipError_t e = shape.halve && !fused && p.B >= 2
                     ? hipErrorOutOfMemory
                     : stage_windows(p, frames_host, ws, ws_bytes, st, slab_bytes, s, before_projection);
  if (e == hipErrorOutOfMemory) {
    // many parts of large windows (fine resolution): the slabs of all frames do not fit the
    // workspace.  Frames are independent, so they go through in two halves.
    if (p.B < 2 || fused) return hipErrorNotSupported;
    shape.halve = true;
    g_last_split[3] += 1;
    dm_params q = p;
    q.B = p.B / 2;
    const size_t n = (size_t)p.H * p.W, m = (size_t)p.mh * p.mw, h = q.B;
    e = run_window(q, frames_host, depth, value, valid, out, mask, height, nullptr, nullptr, ws,
                   ws_bytes, before_projection, nullptr, s);
    if (e == hipErrorNotSupported) shape.hopeless = 63;     // a half that does not fit or pay
    if (e != hipSuccess) return e;
    q.B = p.B - (int)h;
    e = run_window(q, frames_host + h, depth + h * p.dc * n, value ? value + h * p.vc * n : nullptr,
                      valid ? valid + h * p.valid_c * n : nullptr, out + h * oc_total * m,
                      mask + h * oc_total * m, height ? height + h * p.dc * m : nullptr, nullptr,
                      nullptr, ws, ws_bytes, nullptr, after_projection, s);
    if (e == hipErrorNotSupported) shape.hopeless = 63;
    return e;
  }
  if (e != hipSuccess) return e;
#ifdef DM_HOST_TIMING
  const auto t_1 = clk::now();
#endif
  unsigned char* base = static_cast<unsigned char*>(ws);
  float* slabs = reinterpret_cast<float*>(base + st.geom_bytes);

  const bool is_max = p.reduction == DM_REDUCE_MAX;
  e = window_pass(p, st, slabs, depth, value, valid, out, mask, oc_total, p.fill, is_max,
                  slab_bytes, s);
  if (e != hipSuccess) return e;
  if (height && value) {      // maps.py:332-350: second projection, NINF fill, max
    // its mask is not returned (maps.py:340): it goes to scratch at the workspace tail
    const size_t hm = (size_t)p.B * p.dc * p.mh * p.mw;
    if (slab_bytes < hm + (size_t)p.B * st.nparts * st.slab_stride * 4) return hipErrorNotSupported;
    uint8_t* scratch_mask = base + ws_bytes - hm;
    e = window_pass(p, st, slabs, depth, nullptr, valid, height, scratch_mask, p.dc, -INFINITY, true,
                    slab_bytes - hm, s);
    if (e != hipSuccess) return e;
  }
  if (after_projection) {
    e = hipEventRecord(after_projection, s);
    if (e != hipSuccess) return e;
  }
#ifdef DM_HOST_TIMING
  const auto t_2 = clk::now();
#endif
  if (fused) {
    FuseArgs fa;
    fa.B = p.B; fa.b0 = 0; fa.accumulate = 0;
    fa.dc = oc_total; fa.mh = p.mh; fa.mw = p.mw; fa.fill = p.fill;
    fa.unions = st.g_unions; fa.maps = out; fa.fused = fused; fa.fused_mask = fused_mask;
    const dim3 g((unsigned)(((size_t)p.mh * p.mw / 4 + kFuseGroups - 1) / kFuseGroups), oc_total);
    const dim3 blk(kFuseGroups * kFuseLanes);
    e = is_max ? launch(k_fuse_unions<true>, g, blk, 0, s, fa)
               : launch(k_fuse_unions<false>, g, blk, 0, s, fa);
    if (e != hipSuccess) return e;
  }
#ifdef DM_HOST_TIMING
  const auto t_3 = clk::now();
  acc_t[0] += std::chrono::duration<double, std::micro>(t_1 - t_0).count();
  acc_t[1] += std::chrono::duration<double, std::micro>(t_2 - t_1).count();
  acc_t[2] += std::chrono::duration<double, std::micro>(t_3 - t_2).count();
  if (++calls_t % 200 == 0) {
    fprintf(stderr, "[dm timing] stage %.2f us  window_pass %.2f us  fuse %.2f us per call\n",
            acc_t[0] / 200, acc_t[1] / 200, acc_t[2] / 200);
    acc_t[0] = acc_t[1] = acc_t[2] = 0;
  }
#endif
  return hipSuccess;
}

// dm_orth_project_fused_f32: scatter into the LDS windows, then reduce the slabs of
// all frames straight into ONE (oc, mh, mw) map (optionally on top of its content).
hipError_t run_window_fused(const dm_params& p, const dm_frame* frames_host, const float* depth,
                            const float* value, const uint8_t* valid, float* out, uint8_t* mask,
                            int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  if (reinterpret_cast<uintptr_t>(out) % 16 != 0 || reinterpret_cast<uintptr_t>(mask) % 4 != 0)
    return hipErrorNotSupported;
  Staged st;
  size_t slab_bytes = 0;
  hipError_t e = stage_windows(p, frames_host, ws, ws_bytes, st, slab_bytes, s, nullptr, true);
  if (e != hipSuccess) return e;
  return window_pass(p, st, reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + st.geom_bytes),
                     depth, value, valid, nullptr, nullptr,
                     p.vc ? p.vc : p.dc, p.fill, p.reduction == DM_REDUCE_MAX, slab_bytes, s, out,
                     mask, accumulate);
}

}  // namespace dm

extern "C" __attribute__((visibility("default"))) void dm_debug_last_split(int32_t* out4) {
  for (int i = 0; i < 4; ++i) out4[i] = dm::g_last_split[i];
  dm::g_last_split[3] = 0;
}

extern "C" __attribute__((visibility("default"))) int dm_debug_force_bands(int on) {
  const int old = dm::g_force_bands;
  dm::g_force_bands = on != 0;
  return old;
}

#ifdef DM_STAMPS
extern "C" __attribute__((visibility("default"))) void dm_debug_stamp_buffer(long long* dev) {
  dm::g_stamp_buffer = dev;
}
#endif
