// Device side of the strip path: k_strip_scatter and k_strip_merge (dm_strip.hip launches
// them).  Same pixel arithmetic and LDS-window scatter as k_window_scatter
// (dm_window_kernels.hpp), but
//   * the geometry (windows, cone edges, per-row covers: dm_strip_geometry.hpp) is derived ON
//     THE DEVICE from the frame records -- nothing of a call is computed on the host, so the
//     launch sequence depends only on pointers and call-wide constants (graph capturable), and
//   * every float4 group of a strip's window that no other strip can reach is written
//     straight from LDS to the map; only groups two or more strips can reach go through a
//     slab, and k_strip_merge visits only those and the never-reached groups of the union
//     window.
#pragma once

#include "dm_strip_geometry.hpp"
#include "dm_window_kernels.hpp"

namespace dm {
namespace {

constexpr int kGeomBytes = 512;           // LDS reserved for the FrameGeom in front of the cover table
static_assert(sizeof(strip::FrameGeom) == 336 && sizeof(strip::FrameGeom) <= kGeomBytes, "FrameGeom layout (tests/test_hip_strip.py reads it) and its LDS slot");

struct StripArgs {
  int W, H;
  int clip, flip_h;
  float cx, cy, fx, fy, res;
  float fx_inv, fy_inv, res_inv;
  float dmin, dmax, hmax;
  float Hm1, mhm1;
  int wp, P;
  int dc, valid_c;
  int oc, ch0, oc_total;      // channels of this launch's group / first channel / channels of out
  int slab_stride;            // cells per slab = cells of the LDS window region
  int table_off;              // float index in LDS of the FrameGeom (the cover table follows)
  int max_rows;               // rows of a frame's cover table (>= the union window's height)
  float fill;
  int b0;
  const float* frames;        // (B, 32) dm_frame records in device memory
  const float* depth;
  const float* value;         // (B, oc_total, H, W) or NULL: project the heights
  const uint8_t* valid;
  float* slabs;
  float* out;
  uint8_t* mask;
  int mh, mw;
  Win16* g_wins;              // (B, kMaxStrips)     published for k_strip_merge
  Win16* g_unions;            // (B)                 ... and the batch fuse
  strip::RowEntry* g_rows;    // (B, max_rows, P)   per row and strip: cover, owned
  int* status;                // set non-zero when a frame's geometry does not fit the launch
#ifdef DM_STAMPS
  long long* stamps;
#endif
  const strip::Cfg* cfg;      // device copy (in front of the frame records): read by wave 0 only
};

// L1 geometry of one frame by ONE wave: lane s < kMaxStrips derives strip s (the same calls as
// strip::frame_geometry, the host's serial version), the union window is reduced over the lanes.
// y0..ho: the frame record's yaw entries, translation and offsets.
__device__ inline void strip_geometry_wave(const strip::Cfg& c, float y0, float y2, float y6, float y8,
                                           float tx, float tz, float wo, float ho, int lane,
                                           strip::FrameGeom* g) {
  using namespace strip;
  const int s = lane < kMaxStrips ? lane : kMaxStrips - 1;
  float cx[8], cz[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { cx[k] = c.cxl[s][k]; cz[k] = c.czl[s][k]; }
  const float tmin = c.tmin[s], tmax = c.tmax[s];
  const bool live = (lane < c.P) & (c.live[s] != 0);
  const Pose p = pose_of(c, y0, y2, y6, y8, tx, tz, wo, ho);
  Win16 w;
  Line L, R;
  strip_geometry(c, p, cx, cz, tmin, tmax, live, w, L, R);
  if (lane < kMaxStrips) { g->win[lane] = w; g->L[lane] = L; g->R[lane] = R; }
  const bool some = (lane < kMaxStrips) & (w.w > 0);
  int ux0 = some ? w.x0 : 32767, ux1 = some ? w.x0 + w.w : 0;
  int uz0 = some ? w.z0 : 32767, uz1 = some ? w.z0 + w.h : 0;
#pragma unroll
  for (int m = 1; m < kMaxStrips; m <<= 1) {
    ux0 = min(ux0, __shfl_xor(ux0, m, 64)); ux1 = max(ux1, __shfl_xor(ux1, m, 64));
    uz0 = min(uz0, __shfl_xor(uz0, m, 64)); uz1 = max(uz1, __shfl_xor(uz1, m, 64));
  }
  if (lane == 0) {
    g->U = ux1 > ux0 ? Win16{(short)ux0, (short)uz0, (short)(ux1 - ux0), (short)(uz1 - uz0)}
                     : Win16{0, 0, 0, 0};
    g->ok = p.ok;
    g->pad = 0;
  }
}

// RED: kMin / kMax.  Always the fast geometry (axis-aligned rotations, exact FMA division),
// 16-byte depth loads.  HAS_VALUE / HAS_VALID / LEAN as in k_window_scatter.
template <int RED, bool HAS_VALID, bool HAS_VALUE, bool LEAN>
__global__ void __launch_bounds__(kScatterThreads)
k_strip_scatter(StripArgs a) {
  constexpr int VEC = 4;
  // rows of a thread in flight per pipeline stage: value maps carry a second float4 per row and
  // spill at four (the kernel is capped at 128 VGPRs by its 1024 threads)
  constexpr int kRowsInFlight = HAS_VALUE ? 2 : dm::kRowsInFlight;
  extern __shared__ float lds[];
  const int part = blockIdx.x;                 // column strip
  const int chl = blockIdx.y;                  // channel within this launch's group
  const int bl = blockIdx.z, b = a.b0 + bl;
  const int ch = a.ch0 + chl;
  const int dch = a.dc == 1 ? 0 : ch;
  const int nparts = a.P;
  // the strip's pixel rectangle and the first depth rows: kernel arguments only
  const int q0 = part * a.wp;
  int q1 = q0 + a.wp; if (q1 > a.W) q1 = a.W;
  const int r0 = 0, r1 = a.H;
  const int nx = q1 > q0 ? (q1 - q0 + VEC - 1) / VEC : 0;      // (0: a strip past the image's right edge)
  const int ntx = nx < 1 ? 1 : (nx < kScatterThreads ? nx : kScatterThreads);
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
      int rr = r + u * rows_per_iter;
      rr = rr < r1 ? rr : r1 - 1;              // tail rows repeat the last row (max / min: idempotent)
      const float4 t = *reinterpret_cast<const float4*>(dimg + (size_t)rr * a.W + q);
      z[u][0] = t.x; z[u][1] = t.y; z[u][2] = t.z; z[u][3] = t.w;
      if (HAS_VALID) {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          z[u][k] = vimg[(size_t)rr * a.W + q + k] ? z[u][k] : qnan;
      }
      if (HAS_VALUE) {
        const float4 s = *reinterpret_cast<const float4*>(simg + (size_t)rr * a.W + q);
        sv[u][0] = s.x; sv[u][1] = s.y; sv[u][2] = s.z; sv[u][3] = s.w;
      }
    }
  };
  bool first_rows_loaded = false;
  if (gx < nx && q1 > q0) {
    load_rows_at(za, va, q0 + gx * VEC, r0 + gy);
    first_rows_loaded = true;
  }

  // what the pixel loop needs of the frame record: one batch of scalar loads, pinned (the rest
  // of the record is read by wave 0 only, for the geometry)
  const float* tf = a.frames + (size_t)b * 32;
  float p4 = tf[4], p5 = tf[5], p7 = tf[7], p8 = tf[8], cam_h = tf[9];
  float fy0 = tf[10], fy2 = tf[12], fy6 = tf[16], fy8 = tf[18], ftx = tf[19], ftz = tf[20];
  float wo = tf[21], ho = tf[22];
  asm volatile("" : "+s"(p4), "+s"(p5), "+s"(p7), "+s"(p8), "+s"(cam_h), "+s"(fy0), "+s"(fy2),
                    "+s"(fy6), "+s"(fy8), "+s"(ftx), "+s"(ftz), "+s"(wo), "+s"(ho));
#ifdef DM_STAMPS
  long long stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DM_STAMP(0);
  strip::FrameGeom* geom = reinterpret_cast<strip::FrameGeom*>(lds + a.table_off);
  strip::RowEntry* rows = reinterpret_cast<strip::RowEntry*>(lds + a.table_off + kGeomBytes / 4);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const float lds_init = a.fill;
  // Wave 0 derives the frame's geometry while the others initialise the whole window region
  // of LDS (the window's size is not known before the geometry is).
  if (wave == 0)
    strip_geometry_wave(*a.cfg, fy0, fy2, fy6, fy8, ftx, ftz, wo, ho, (int)threadIdx.x & 63, geom);
  DM_STAMP(1);
  for (int i = threadIdx.x * 4; i < a.slab_stride + 64; i += kScatterThreads * 4)
    *reinterpret_cast<float4*>(lds + i) = make_float4(lds_init, lds_init, lds_init, lds_init);
  lds_barrier();
  DM_STAMP(2);
  Window w = widen(geom->win[part]);
  Window U = widen(geom->U);
  {   // the union window in whole 128-byte lines (covers reach that far)
    const int ux1 = min((U.x0 + U.w + strip::kSpanAlign - 1) & ~(strip::kSpanAlign - 1), a.mw);
    U.x0 &= ~(strip::kSpanAlign - 1);
    U.w = U.w > 0 ? ux1 - U.x0 : 0;
  }
  {
    // wave-uniform values: keep them in SGPRs
    w.x0 = __builtin_amdgcn_readfirstlane(w.x0); w.z0 = __builtin_amdgcn_readfirstlane(w.z0);
    w.w = __builtin_amdgcn_readfirstlane(w.w); w.h = __builtin_amdgcn_readfirstlane(w.h);
    U.x0 = __builtin_amdgcn_readfirstlane(U.x0); U.z0 = __builtin_amdgcn_readfirstlane(U.z0);
    U.w = __builtin_amdgcn_readfirstlane(U.w); U.h = __builtin_amdgcn_readfirstlane(U.h);
  }
  // a frame that does not fit what the host sized the launch for: flag it and project nothing
  // (cannot happen when the host derived the sizes from these very frames)
  if (w.w * w.h > a.slab_stride || U.h > a.max_rows || !__builtin_amdgcn_readfirstlane(geom->ok)) {
    if (threadIdx.x == 0 && (U.w > 0 || !geom->ok)) atomicOr(a.status, 1);
    w = Window{0, 0, 0, 0};
    U = Window{0, 0, 0, 0};
  }
  const int area = w.w * w.h;
  // Row table of the frame (cover and owned span of every strip on every row of the union
  // window): built right before the pixel loop, under the first depth loads in flight, by all
  // threads -- thread = (row, strip), the strips of a row in neighbouring lanes, which
  // exchange their covers by shuffles.  Read back only behind the barrier that ends the loop.
  const bool publisher = part == 0 && chl == 0;
  auto build_rows = [&]() {
    const int P2 = nparts <= 4 ? 4 : 8;
    const int sub = (int)threadIdx.x & (P2 - 1);
    const int per_pass = kScatterThreads / P2;
    for (int r = (int)threadIdx.x / P2; r < U.h; r += per_pass) {
      const int ps = sub < nparts ? sub : 0;
      uint32_t cover = strip::row_cover(geom->win[ps], geom->L[ps], geom->R[ps], U.z0 + r, a.mw);
      cover = sub < nparts ? cover : 0u;
      int lo = (int)(cover & 0xffffu), hi = (int)(cover >> 16);
      for (int m = 1; m < P2; ++m) strip::cut_span(lo, hi, (uint32_t)__shfl_xor((int)cover, m, 64));
      const strip::RowEntry e = {cover, hi > lo ? (uint32_t)lo | ((uint32_t)hi << 16) : 0u};
      if (sub < nparts) {
        rows[r * nparts + sub] = e;
        if (publisher) a.g_rows[((size_t)b * a.max_rows + r) * nparts + sub] = e;
      }
    }
  };
  DM_STAMP(3);
  // Fill duty (as in k_window_scatter): map rows part, part + P, ... outside the union window
  const int g4 = a.mw >> 2;
  const int fill_rows = (a.mh - part + nparts - 1) / nparts;
  const size_t map_base = ((size_t)b * a.oc_total + ch) * (size_t)a.mh * a.mw;
  const bool do_fill = a.out != nullptr && fill_rows > 0;
  const int fill_total = do_fill ? fill_rows * g4 : 0;
  const int fill_steps = (fill_total + kScatterThreads - 1) / kScatterThreads;
  const float g4_inv = 1.0f / (float)g4;
  int fs = 0;
  // Buffer stores: the map of this (frame, channel) as a raw buffer resource.  A store that has
  // nothing to write gets an offset past the end of the buffer and is dropped by the hardware's
  // range check: no branch, no dummy destination, scalar base + 32-bit offsets instead of 64-bit
  // address arithmetic (-1 us).  (Write-through stores, sc0 sc1, measured the same as plain
  // ones here, for this kernel and for the kernel boundary behind it.)
  constexpr int kFillCachePolicy = 0;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const unsigned map_cells = (unsigned)a.mh * (unsigned)a.mw;
  const __amdgpu_buffer_rsrc_t rs_out =
      __builtin_amdgcn_make_buffer_rsrc(a.out + map_base, 0, do_fill ? map_cells * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mask =
      __builtin_amdgcn_make_buffer_rsrc(a.mask + map_base, 0, do_fill ? map_cells : 0u, 0x00020000);
  const unsigned fill_bits = __float_as_uint(a.fill);
  auto fill_step = [&]() {
    const int i = fs * kScatterThreads + (int)threadIdx.x;
    ++fs;
    int k = (int)((float)i * g4_inv);
    k -= (k * g4 > i);
    k += ((k + 1) * g4 <= i);
    const int g = i - k * g4;
    const int r = part + k * nparts, x = g << 2;
    const bool skip = (i >= fill_total) | (((unsigned)(r - U.z0) < (unsigned)U.h) &
                                           ((unsigned)(x - U.x0) < (unsigned)U.w));   // (no branch)
    const int cell = r * a.mw + x;
    const int ob = skip ? 0x7ffffff0 : cell * 4, mb = skip ? 0x7ffffff0 : cell;
    __builtin_amdgcn_raw_buffer_store_b128((u32x4){fill_bits, fill_bits, fill_bits, fill_bits}, rs_out, ob, 0,
                                           kFillCachePolicy);
    __builtin_amdgcn_raw_buffer_store_b32(0u, rs_mask, mb, 0, kFillCachePolicy);
  };
  // the reductions' identities as float4s in front of the slabs (k_strip_merge reads them where
  // a strip has nothing for a group)
  if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x < 8)
    a.slabs[(int)threadIdx.x - 8] = threadIdx.x < 4 ? -INFINITY : INFINITY;
  auto publish_geometry = [&]() {
    if (publisher && threadIdx.x < strip::kMaxStrips)
      a.g_wins[(size_t)b * strip::kMaxStrips + threadIdx.x] = geom->win[threadIdx.x];
    if (publisher && threadIdx.x == 0) a.g_unions[b] = narrow16(U);
  };

  // (a local map's records carry a neutral yaw and no translation: dm_strip.hip stage_frames)
  const float y0 = fy0, y2r = fy2, y6 = fy6, y8 = fy8, tx = ftx, tz = ftz;
  const float flip_s = a.flip_h ? -1.0f : 1.0f, flip_c = a.flip_h ? a.mhm1 : 0.0f;
  const unsigned dummy = (unsigned)a.slab_stride + (threadIdx.x & 63u);   // 64 scratch cells

  if (area > 0) {
    for (int g = gx; g < nx; g += ntx) {       // one trip unless the strip is wider than the block
      const int q = q0 + g * VEC;
      float ax[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float d = (float)(q + k) - a.cx;
        ax[k] = div_markstein(d, a.fx, a.fx_inv);
        if (!LEAN) ax[k] = (q + k < a.clip || q + k >= a.W - a.clip) ? qnan : ax[k];
      }
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
          rr = rr < r1 ? rr : r1 - 1;
          float yr = (float)rr;
          yr = a.flip_h ? a.Hm1 - yr : yr;                       // maps.py:670-671
          const float dy = yr - a.cy;
          float ay = div_markstein(dy, a.fy, a.fy_inv);
          if (!LEAN) ay = (rr < a.clip || rr >= a.H - a.clip) ? qnan : ay;
          unsigned li[VEC];
          float hv[VEC];
          float xfv[VEC], zfv[VEC], h1v[VEC];
          typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int k = 0; k < VEC; k += 2) {
            const f2 zz = {z[u][k], z[u][k + 1]};
            const f2 axp = {ax[k], ax[k + 1]};
            const f2 X = axp * zz;
            const f2 Y = zz * ay;                                            // maps.py:677-678
            const f2 h1 = __builtin_elementwise_fma(zz, (f2){p7, p7}, Y * p4) + cam_h;   // maps.py:790-797
            const f2 z1 = __builtin_elementwise_fma(zz, (f2){p8, p8}, Y * p5);
            const f2 x2 = __builtin_elementwise_fma(z1, (f2){y6, y6}, X * y0) + tx;      // maps.py:884-892
            const f2 z2 = __builtin_elementwise_fma(z1, (f2){y8, y8}, X * y2r) + tz;
            const f2 ri = {a.res_inv, a.res_inv}, nres = {-a.res, -a.res};
            const f2 qx = x2 * ri, qz = z2 * ri;                             // exact division
            f2 xf2 = __builtin_elementwise_fma(__builtin_elementwise_fma(nres, qx, x2), ri, qx) + wo;
            f2 zf2 = __builtin_elementwise_fma(__builtin_elementwise_fma(nres, qz, z2), ri, qz) + ho;
            zf2 = __builtin_elementwise_fma(zf2, (f2){flip_s, flip_s}, (f2){flip_c, flip_c});
            xf2 = xf2 + 0.5f;
            zf2 = zf2 + 0.5f;
            xfv[k] = xf2.x; xfv[k + 1] = xf2.y;
            zfv[k] = zf2.x; zfv[k + 1] = zf2.y;
            h1v[k] = h1.x; h1v[k + 1] = h1.y;
          }
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const float zz = z[u][k], xf = xfv[k], zf = zfv[k], h1 = h1v[k];
            // maps.py:537-544, 286-288, 1150-1158
            const unsigned ux = (unsigned)(floor_to_int(xf) - w.x0);
            const unsigned uz = (unsigned)(floor_to_int(zf) - w.z0);
            // (bitwise &: the short-circuit form compiles to a branch per pixel, and a branch in
            // this loop costs its counted waits)
            bool ok = (ux < (unsigned)w.w) & (uz < (unsigned)w.h) & (zz >= a.dmin) & (zz <= a.dmax);
            if (!LEAN) ok = ok & !__builtin_isunordered(xf, zf) & (h1 <= a.hmax);
            const float sval = HAS_VALUE ? sv[u][k] : h1;
            if (HAS_VALUE) ok = ok & (sval == sval);             // NaN never replaces a number
            unsigned cell = __umul24(uz, (unsigned)w.w) + ux;
            asm("" : "+v"(cell));
            li[k] = ok ? cell : dummy;
            hv[k] = sval;
          }
          if (__builtin_amdgcn_ballot_w64(li[0] == li[VEC - 1] && li[0] != dummy) != 0) {
#pragma unroll
            for (int k = 0; k + 1 < VEC; ++k) {
              const bool same = li[k] == li[k + 1];
              const float m = combine<RED>(hv[k], hv[k + 1]);
              hv[k + 1] = same ? m : hv[k + 1];
              li[k] = same ? dummy : li[k];
            }
          }
#pragma unroll
          for (int k = 0; k < VEC; ++k) lds_reduce<RED>(lds + li[k], hv[k]);
        }
      };
      const int niter = (r1 - r0 + step - 1) / step;
      auto pipeline = [&](auto with_fill) {
        constexpr bool kFill = decltype(with_fill)::value;
        int r = r0 + gy;
        if (!first_rows_loaded) load_rows(za, va, r);
        first_rows_loaded = false;
        // (the row table is built here, under the first depth rows in flight, and NOT inside the
        // loop: a branch in the loop body makes the compiler drain its counted waits)
        // two groups of rows in flight while the row table is built (VALU + LDS only)
        load_rows(zb_, vb_, r + step);
        DM_STAMP(7);
        if (g == gx) build_rows();
        DM_STAMP(8);
        // younger waves of a SIMD get the higher issue priority in the loop (age arbitration
        // favours the oldest wave otherwise, and the last wave left on a SIMD runs latency
        // bound).  Only now: under static priorities the four waves of a SIMD run the code
        // above one after the other instead of hiding each other's latencies.
        if (wave >= 12) __builtin_amdgcn_s_setprio(3);
        else if (wave >= 8) __builtin_amdgcn_s_setprio(2);
        else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
        for (int it = 0; it < niter; it += 2) {
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
            load_rows(zb_, vb_, r + 3 * step);        // (past the end: the last row again, unused)
          }
          r += 2 * step;
        }
      };
      if (do_fill) pipeline(std::true_type{}); else pipeline(std::false_type{});
    }
  }
  if (area == 0 || nx == 0) build_rows();     // (no pixel loop ran: the table is still owed to k_strip_merge)
  DM_STAMP(4);
  while (fs < fill_steps) fill_step();
  __builtin_amdgcn_s_setprio(0);
  lds_barrier();
  DM_STAMP(5);
  if (area > 0) {
    // Flush: 16 lanes per window row.  The groups of this strip's cover go straight to the map
    // where the strip owns them (no other strip's cover reaches them), else to the slab, which
    // k_strip_merge combines with the other strips'.
    const int pid = (b * a.oc + chl) * nparts + part;
    float* slab = a.slabs + (size_t)pid * a.slab_stride;
    const int l16 = (int)threadIdx.x & 15;
    for (int row = (int)threadIdx.x >> 4; row < w.h; row += kScatterThreads / 16) {
      const int z = w.z0 + row;
      const strip::RowEntry e = rows[(z - U.z0) * nparts + part];
      const int lo = (int)(e.cover & 0xffffu), hi = (int)(e.cover >> 16);
      const int cell0 = row * w.w - w.x0;
      for (int x = lo + (l16 << 2); x < hi; x += 64) {
        // (a span is whole 128-byte lines: outside the window nothing landed)
        const bool inside = (unsigned)(x - w.x0) < (unsigned)w.w;
        float4 v = make_float4(a.fill, a.fill, a.fill, a.fill);
        if (inside) v = *reinterpret_cast<const float4*>(lds + cell0 + x);
        if (strip::in_span(e.owned, x) && a.out != nullptr) {
          const size_t cell = map_base + (size_t)z * a.mw + x;
          *reinterpret_cast<float4*>(a.out + cell) = v;
          const uint32_t mk = (uint32_t)mask_of(v.x, a.fill) | ((uint32_t)mask_of(v.y, a.fill) << 8) |
                              ((uint32_t)mask_of(v.z, a.fill) << 16) |
                              ((uint32_t)mask_of(v.w, a.fill) << 24);
          *reinterpret_cast<uint32_t*>(a.mask + cell) = mk;
        } else if (inside) {
          *reinterpret_cast<float4*>(slab + cell0 + x) = v;
        }
      }
    }
  }
  publish_geometry();
  DM_STAMP(6);
  DM_STAMPS_OUT();
}

struct StripMergeArgs {
  int b0, oc, ch0, oc_total, mh, mw;
  int P, slab_stride, max_rows;
  float fill;
  const Win16* g_wins;
  const Win16* g_unions;
  const strip::RowEntry* g_rows;
  const float* slabs;         // (the 8 floats in front of it: -inf x 4, +inf x 4)
  float* out;
  uint8_t* mask;
};

constexpr int kMergeRowsPerWave = 4;
constexpr int kMergeRowsPerBlock = kMergeThreads / 64 * kMergeRowsPerWave;
static_assert(strip::kSpanAlign == 4, "k_strip_merge reads a slab wherever a cover is: covers must lie inside the windows");

// One wave per kMergeRowsPerWave map rows of a frame's union window, lanes along the row: a
// group inside some strip's owned span was written by k_strip_scatter; every other group gets
// the max / min of the slabs of the strips whose covers hold it (the fill value where none
// does).  The kernel is bound by instruction issue (16 K rows of a few groups each), so what a
// row needs is kept wave-uniform -- lane p fetches strip p's row entry and window in one batch
// of loads, the values are broadcast from the lanes, the span arithmetic runs on the scalar
// unit -- and P is a template parameter: P slab loads in flight per row, no loop.  Strips with
// nothing for a group load `ident` (a float4 of the reduction's identity) instead.
template <int RED, int P>
__global__ void __launch_bounds__(kMergeThreads)
k_strip_merge(StripMergeArgs a) {
  const int fcl = blockIdx.y;                  // (frame of the launch) * oc + channel of the group
  const int bl = fcl / a.oc, b = a.b0 + bl;
  const int chl = fcl - bl * a.oc;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  const int row0 = (blockIdx.x * (kMergeThreads / 64) + wave) * kMergeRowsPerWave;
  if (row0 >= a.max_rows) return;
  const int pi = lane < P ? lane : P - 1;
  const Win16 wv = a.g_wins[(size_t)b * strip::kMaxStrips + pi];
  const Window U = widen(a.g_unions[b]);
  const strip::RowEntry* re = a.g_rows + ((size_t)b * a.max_rows + row0) * P + pi;
  strip::RowEntry ev[kMergeRowsPerWave];
#pragma unroll
  for (int k = 0; k < kMergeRowsPerWave; ++k)   // (rows past max_rows: clamped, unused)
    ev[k] = re[(size_t)(row0 + k < a.max_rows ? k : 0) * P];
  int wxz = (int)(uint16_t)wv.x0 | ((int)(uint16_t)wv.z0 << 16), www = (int)(uint16_t)wv.w;
  // The lanes' values are read across lanes inside divergent code below: they must exist in
  // EVERY lane, i.e. be loaded here, where all lanes are active (left to itself the compiler
  // sinks the loads to their uses, where only the lanes of the row's groups execute them).
  asm volatile("" : "+v"(wxz), "+v"(www));
#pragma unroll
  for (int k = 0; k < kMergeRowsPerWave; ++k) asm volatile("" : "+v"(ev[k].cover), "+v"(ev[k].owned));
  // Lane p prepares strip p's numbers for every row (vector ALU, all strips at once: the scalar
  // unit is shared by the CU's waves and would be the bottleneck of this kernel); they are
  // broadcast from the lanes where they are used.
  //   base: where cell (0, 0) of the map would lie in the strip's slab, as a 32-bit float index
  //         relative to `slabs` (the slabs of one channel group hold fewer than 2^31 floats)
  const int ww_l = www;
  const int base_l = pi * a.slab_stride - (wxz >> 16) * ww_l - (int)(int16_t)(wxz & 0xffff);
  int clo_l[kMergeRowsPerWave], olo_l[kMergeRowsPerWave], ohi_l[kMergeRowsPerWave],
      chi_l[kMergeRowsPerWave], off_l[kMergeRowsPerWave];
#pragma unroll
  for (int k = 0; k < kMergeRowsPerWave; ++k) {
    const uint32_t cover = ev[k].cover, owned = ev[k].owned;
    clo_l[k] = (int)(cover & 0xffffu); chi_l[k] = (int)(cover >> 16);
    olo_l[k] = owned ? (int)(owned & 0xffffu) : clo_l[k];      // (no owned span: an empty one at clo)
    ohi_l[k] = owned ? (int)(owned >> 16) : clo_l[k];
    off_l[k] = base_l + (U.z0 + row0 + k) * ww_l;
    // (read across lanes inside divergent code below: must be computed here, in every lane)
    asm volatile("" : "+v"(clo_l[k]), "+v"(olo_l[k]), "+v"(ohi_l[k]), "+v"(chi_l[k]), "+v"(off_l[k]));
  }
  const size_t fo = (size_t)b * a.oc_total + a.ch0 + chl;
  const float* slabs = a.slabs + ((size_t)(b * a.oc + chl) * P) * a.slab_stride;
  // the reduction's identity as a float4 in memory, 8 (max) / 4 (min) floats in front of the slabs
  const int ident_at = (int)(a.slabs - slabs) - (RED == kMax ? 8 : 4);
  const float ident = RED == kMax ? -INFINITY : INFINITY;
#pragma unroll
  for (int k = 0; k < kMergeRowsPerWave; ++k) {
    const int row = row0 + k;
    if (row >= U.h) break;                     // wave-uniform
    const int zb = U.z0 + row;
    float* const out_row = a.out + fo * (size_t)a.mh * a.mw + (size_t)zb * a.mw;
    uint8_t* const mask_row = a.mask + fo * (size_t)a.mh * a.mw + (size_t)zb * a.mw;
    // per strip: the two pieces of its cover around its owned span, its slab row's offset
    int clo[P], olo[P], ohi[P], chi[P], off[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      clo[p] = __builtin_amdgcn_readlane(clo_l[k], p); olo[p] = __builtin_amdgcn_readlane(olo_l[k], p);
      ohi[p] = __builtin_amdgcn_readlane(ohi_l[k], p); chi[p] = __builtin_amdgcn_readlane(chi_l[k], p);
      off[p] = __builtin_amdgcn_readlane(off_l[k], p);
    }
    for (int x = U.x0 + (lane << 2); x < U.x0 + U.w; x += 256) {
      bool is_owned = false;
      float4 sv[P];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        is_owned = is_owned | ((x >= olo[p]) & (x < ohi[p]));
        const bool hit = ((x >= clo[p]) & (x < olo[p])) | ((x >= ohi[p]) & (x < chi[p]));
        const int at = hit ? off[p] + x : ident_at;
        sv[p] = *reinterpret_cast<const float4*>(slabs + at);
      }
      float4 acc = make_float4(ident, ident, ident, ident);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        acc.x = combine<RED>(acc.x, sv[p].x);
        acc.y = combine<RED>(acc.y, sv[p].y);
        acc.z = combine<RED>(acc.z, sv[p].z);
        acc.w = combine<RED>(acc.w, sv[p].w);
      }
      // (the fill value takes part: utils.py:470-477 reduces INTO the filled canvas)
      acc.x = combine<RED>(acc.x, a.fill); acc.y = combine<RED>(acc.y, a.fill);
      acc.z = combine<RED>(acc.z, a.fill); acc.w = combine<RED>(acc.w, a.fill);
      if (is_owned) continue;
      *reinterpret_cast<float4*>(out_row + x) = acc;
      const uint32_t mk = (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
                          ((uint32_t)mask_of(acc.z, a.fill) << 16) |
                          ((uint32_t)mask_of(acc.w, a.fill) << 24);
      *reinterpret_cast<uint32_t*>(mask_row + x) = mk;
    }
  }
}

// Test hook kernel: the geometry of every frame exactly as k_strip_scatter derives it.
__global__ void __launch_bounds__(64)
k_strip_geometry_dump(const strip::Cfg* cfg, const float* frames, strip::FrameGeom* out) {
  __shared__ strip::FrameGeom g;
  const float* f = frames + (size_t)blockIdx.x * 32;
  strip_geometry_wave(*cfg, f[10], f[12], f[16], f[18], f[19], f[20], f[21], f[22], (int)threadIdx.x, &g);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = g;
}

}  // namespace
}  // namespace dm
