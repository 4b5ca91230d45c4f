// Geometry of the strip path (dm_strip.hip): which map cells each column strip of a frame can
// reach, evaluated by the SAME code on the device (inside k_strip_scatter / k_strip_merge) and
// on the host (sizing, eligibility, the CPU tests through dm_debug_strip_geometry).
//
// A frame is cut into P column strips (pr = pd = 1).  With the projector's axis-aligned
// rotations (pitch about X, yaw about Y) the map position of a pixel with ray slopes (ax, ay)
// and depth d is
//     c = A + d * (ax * U + g(ay) * F),     g(ay) = Rp[5] * ay + Rp[8],
// A = the camera's cell, U / F = the camera's right / forward direction in cells per metre.
// So every pixel of a strip lies in the CONE  { A + b * (t * U + F) : b >= 0, t in [tmin, tmax] }
// with t = ax / g over the strip's corner rays, between its near and far depth.  From that:
//
//   window  W_p   bounding box of the truncated cone (+ slack), x aligned to 4: the strip's
//                 LDS image.
//   cover   C_p(z) for a map row z: the cells of W_p's row z inside the cone widened by the
//                 slack, as [lo, hi) aligned to 4 -- a superset of what strip p can hit.
//
// A float4 group of a map row is OWNED by strip p when it lies in C_p(z) and in no other
// strip's cover: p then writes it straight to the map.  Groups in two or more covers go
// through slabs and k_strip_merge; groups of the frame's union window in no cover hold the
// fill value (k_strip_merge writes them).  All of that only needs every consumer to compute
// identical covers -- this file -- and the covers to be supersets, which
// tests/test_strip_geometry.py checks against per-pixel cells computed on the CPU.
//
// Float32, no libm calls beyond floorf / ceilf (IEEE +,-,*,/ with -ffp-contract=off and
// correctly rounded division on both sides): identical results on host and device.
#pragma once

#include <math.h>
#include <stdint.h>

#include "dm_window_geometry.hpp"

namespace dm {
namespace strip {

constexpr int kMaxStrips = 8;
constexpr int kSpanAlign = 4;            // cells: covers / owned spans are whole float4 groups of the map

// Everything of the geometry that does not depend on the camera's yaw and position: the call's
// parameters plus the batch's camera pitch (dm_strip.hip requires one pitch per batch and
// builds this once per camera rig).  Lives in device memory in front of the frame records.
struct Cfg {
  int P;                        // column strips
  int mw, mh;
  int flip_h;
  int cone_ok;                  // every image row looks forward (g > 0): the cone model holds
  float inv;                    // cells per metre (1 / map_res)
  float reach;                  // bound of a point's distance from the camera in cells
  float pad;
  int live[kMaxStrips];         // 0: nothing left of the strip after clip_border
  float tmin[kMaxStrips];       // cone edges: t = ax / g over the strip's corner rays
  float tmax[kMaxStrips];
  // corners of the strip's truncated cone in the camera's local frame (x right, z forward), in
  // cells: index bit 0 = near/far, bit 1 = ax lo/hi, bit 2 = ay lo/hi
  float cxl[kMaxStrips][8];
  float czl[kMaxStrips][8];
};

// The same pose-independent inputs as the kernels take them -- by value, in their kernel
// arguments, so that a launch needs no table in device memory: Cfg without its corner tables,
// which every lane derives for its strip with the host's expressions (strip_corners).
struct RigArgs {
  int P, mw, mh, flip_h, cone_ok;
  int live_mask;                // bit s: strip s has pixels left after clip_border
  float inv, reach;
  float g0, g1;                 // z1 = g * depth at the extreme live rows (cfg_rig)
  float dmin, dmax;
  float ax_lo[kMaxStrips], ax_hi[kMaxStrips];     // ray slopes of each strip's first / last live column
  float tmin[kMaxStrips], tmax[kMaxStrips];
};

// Corners of a strip's truncated cone in the camera's local frame, in cells (index bit 0 =
// near/far, bit 1 = ax lo/hi, bit 2 = ay lo/hi): one expression for host and device.
__host__ __device__ inline void strip_corners(float ax_lo, float ax_hi, float g0, float g1, float dmin,
                                              float dmax, float inv, float* cx, float* cz) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float d = (k & 1) ? dmax : dmin;
    const float ax = (k & 2) ? ax_hi : ax_lo;
    const float g = (k & 4) ? g1 : g0;
    cx[k] = ax * d * inv;
    cz[k] = g * d * inv;
  }
}

// One frame's geometry (L1): windows and cone edges of every strip.
// Edge lines in cell coordinates: a cell centre (x, z) is inside the widened cone of strip p
// iff  L.nx * x + L.nz * z + L.k >= 0  and  R.nx * x + R.nz * z + R.k >= 0.
struct Line { float nx, nz, k, inv_nx; };
struct FrameGeom {
  Win16 win[kMaxStrips];
  Win16 U;                      // bounding box of the windows, x aligned to 4
  Line L[kMaxStrips], R[kMaxStrips];
  int ok;                       // 0: the cone model does not apply to this frame
  int inside;                   // bit s: strip s's window was not clipped by the map's borders, i.e.
                                // EVERY pixel of the strip with a depth in range lands inside it
};

// Per (map row of the union window, strip): what the strip can reach on that row and the part
// of it nobody else can -- both [lo, hi) in cells, multiples of 4, packed lo | hi << 16
// (0 = empty).  owned is a sub-interval of cover that no other strip's cover intersects.
struct RowEntry { uint32_t cover, owned; };

__host__ __device__ inline float fabs_(float v) { return v < 0.0f ? -v : v; }
__host__ __device__ inline float fmin2(float a, float b) { return a < b ? a : b; }
__host__ __device__ inline float fmax2(float a, float b) { return a > b ? a : b; }
__host__ __device__ inline bool finite_f(float v) { return v == v && v - v == 0.0f; }

// Host: the pitch-dependent part of Cfg from the pitch rotation Rp (row-major, utils.py:326-327;
// axis-aligned: x1 = X, z1 = Rp[5] * Y + Rp[8] * Z) and the strips' ray slopes.
//   ax_lo / ax_hi: (q - cx) / fx of each strip's first / last live column;  ay_lo / ay_hi: the
//   extreme (y - cy) / fy of the live rows;  depth range [dmin, dmax].
__host__ inline void cfg_rig(Cfg& c, const float* Rp, const float* ax_lo, const float* ax_hi,
                             float ay_lo, float ay_hi, float dmin, float dmax) {
  const float p5 = Rp[5], p8 = Rp[8];
  const float g0 = p5 * ay_lo + p8, g1 = p5 * ay_hi + p8;   // z1 = g * depth
  c.cone_ok = g0 > 1e-3f && g1 > 1e-3f && finite_f(g0) && finite_f(g1);
  const float gmax = fmax2(fabs_(g0), fabs_(g1));
  float amax = 0.0f;
  for (int s = 0; s < kMaxStrips; ++s) {
    c.tmin[s] = c.tmax[s] = 0.0f;
    for (int k = 0; k < 8; ++k) c.cxl[s][k] = c.czl[s][k] = 0.0f;
    if (s >= c.P || !c.live[s]) continue;
    amax = fmax2(amax, fmax2(fabs_(ax_lo[s]), fabs_(ax_hi[s])));
    if (c.cone_ok) {
      const float t0 = ax_lo[s] / g0, t1 = ax_lo[s] / g1, t2 = ax_hi[s] / g0, t3 = ax_hi[s] / g1;
      c.tmin[s] = fmin2(fmin2(t0, t1), fmin2(t2, t3));
      c.tmax[s] = fmax2(fmax2(t0, t1), fmax2(t2, t3));
    }
    strip_corners(ax_lo[s], ax_hi[s], g0, g1, dmin, dmax, c.inv, c.cxl[s], c.czl[s]);
  }
  c.reach = dmax * c.inv * (amax + gmax);
}

// The kernel-argument form of a Cfg (same values; the corners are re-derived on the device).
__host__ inline RigArgs rig_args(const Cfg& c, const float* Rp, const float* ax_lo, const float* ax_hi,
                                 float ay_lo, float ay_hi, float dmin, float dmax) {
  RigArgs r;
  r.P = c.P; r.mw = c.mw; r.mh = c.mh; r.flip_h = c.flip_h; r.cone_ok = c.cone_ok;
  r.live_mask = 0;
  r.inv = c.inv; r.reach = c.reach;
  r.g0 = Rp[5] * ay_lo + Rp[8]; r.g1 = Rp[5] * ay_hi + Rp[8];
  r.dmin = dmin; r.dmax = dmax;
  for (int s = 0; s < kMaxStrips; ++s) {
    const bool on = s < c.P && c.live[s];
    r.live_mask |= on ? 1 << s : 0;
    r.ax_lo[s] = on ? ax_lo[s] : 0.0f; r.ax_hi[s] = on ? ax_hi[s] : 0.0f;
    r.tmin[s] = c.tmin[s]; r.tmax[s] = c.tmax[s];
  }
  return r;
}

// The pose-dependent rest: the yaw rotation's four entries (identity for a local map), the
// camera's cell A = (xd, zd), the slack in cells around everything derived here.  The device
// computes cell coordinates in float32 (error: a few ulp of the largest intermediate), the
// reference does the same float32 operations, so the "true" cell is the float32 one; this
// geometry is float32 as well.  2 cells cover map coordinates up to ~2^17; beyond that the
// slack grows with the magnitude (16 ulp_f32), and past 16 cells the strip path is not used.
struct Pose { float y0, y2, y6, y8, xd, zd, slack; int ok; };

template <class C>
__host__ __device__ inline Pose pose_of(const C& c, float y0, float y2, float y6, float y8, float tx,
                                        float tz, float wo, float ho) {
  Pose p;
  const float fs = c.flip_h ? -1.0f : 1.0f;                 // maps.py:1006-1009: zf -> (mh - 1) - zf
  p.y0 = y0; p.y6 = y6; p.y2 = fs * y2; p.y8 = fs * y8;
  const float txc = tx * c.inv, tzc = tz * c.inv;
  p.xd = txc + wo;
  const float zd = tzc + ho;
  p.zd = c.flip_h ? (float)(c.mh - 1) - zd : zd;
  const float mag = fmax2(fmax2(fabs_(txc), fabs_(tzc)), fmax2(fabs_(wo), fabs_(ho)));
  const float m = mag + c.reach + fabs_(p.xd) + fabs_(p.zd);
  p.slack = 2.0f + 16.0f * m * (1.0f / 8388608.0f);
  p.ok = c.cone_ok && finite_f(p.xd) && finite_f(p.zd) && finite_f(y0) && finite_f(y2) &&
         finite_f(y6) && finite_f(y8) && p.slack <= 16.0f;
  return p;
}

// Window from the bounding box [lx, hx] x [lz, hz] of the eight corners.
template <class C>
__host__ __device__ inline Win16 window_of(const C& c, float lx, float hx, float lz, float hz,
                                           float slack, bool& inside) {
  float x0 = floorf(floorf(lx + 0.5f) - slack), x1 = ceilf(floorf(hx + 0.5f) + slack + 1.0f);
  float z0 = floorf(floorf(lz + 0.5f) - slack), z1 = ceilf(floorf(hz + 0.5f) + slack + 1.0f);
  inside = (x0 >= 0.0f) & (z0 >= 0.0f) & (x1 <= (float)c.mw) & (z1 <= (float)c.mh);   // (false for NaN)
  x0 = x0 < 0.0f ? 0.0f : x0;
  z0 = z0 < 0.0f ? 0.0f : z0;
  x1 = x1 > (float)c.mw ? (float)c.mw : x1;
  z1 = z1 > (float)c.mh ? (float)c.mh : z1;
  const bool some = (x0 < x1) & (z0 < z1);
  const int ix0 = ((int)x0) & ~3;
  int ixe = ((int)x1 + 3) & ~3;                 // mw % 4 == 0 is a precondition
  ixe = ixe > c.mw ? c.mw : ixe;
  return some ? Win16{(short)ix0, (short)(int)z0, (short)(ixe - ix0), (short)((int)z1 - (int)z0)}
              : Win16{0, 0, 0, 0};
}

// Edge line of the cone through A with direction D = t * U + F (U, F: the camera's right /
// forward vectors in cells per metre).  left: the edge at tmin (inside: t >= tmin), else the
// edge at tmax.  The margin (half a cell plus the slack, in the line's own units) is in k.
template <class C>
__host__ __device__ inline Line cone_edge(const C& c, const Pose& p, float t, bool left) {
  const float Dx = (t * p.y0 + p.y6) * c.inv, Dz = (t * p.y2 + p.y8) * c.inv;
  // sgn = sign of cr(F, U) = y6 * y2 - y8 * y0 (orientation: flips with flip_h)
  const float sgn = (p.y6 * p.y2 - p.y8 * p.y0) < 0.0f ? -1.0f : 1.0f;
  Line l;
  l.nx = left ? -sgn * Dz : sgn * Dz;                       // sgn * cr(D, w)  /  sgn * cr(w, D)
  l.nz = left ? sgn * Dx : -sgn * Dx;
  l.k = -l.nx * p.xd - l.nz * p.zd + (0.5f + p.slack) * (fabs_(l.nx) + fabs_(l.nz));
  l.inv_nx = l.nx != 0.0f ? 1.0f / l.nx : 0.0f;
  return l;
}

// One strip's window and cone edges for a pose: the strip's local corners rotated by the yaw.
template <class C>
__host__ __device__ inline void strip_geometry(const C& c, const Pose& p, const float* cxl,
                                               const float* czl, float tmin, float tmax, bool live,
                                               Win16& w, Line& L, Line& R, bool& inside) {
  float lx = INFINITY, hx = -INFINITY, lz = INFINITY, hz = -INFINITY;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float xf = p.y0 * cxl[k] + p.y6 * czl[k] + p.xd;    // maps.py:884-892 in cells
    const float zf = p.y2 * cxl[k] + p.y8 * czl[k] + p.zd;
    lx = fmin2(lx, xf); hx = fmax2(hx, xf); lz = fmin2(lz, zf); hz = fmax2(hz, zf);
  }
  const bool on = live & (p.ok != 0);
  bool in = false;
  const Win16 ww = window_of(c, lx, hx, lz, hz, p.slack, in);
  const Line l = cone_edge(c, p, tmin, true), r = cone_edge(c, p, tmax, false);
  inside = on & in & (ww.w > 0);
  w = on ? ww : Win16{0, 0, 0, 0};
  L = on ? l : Line{0.0f, 0.0f, 0.0f, 0.0f};
  R = on ? r : Line{0.0f, 0.0f, 0.0f, 0.0f};
}

// The whole L1 geometry of one frame, serially (host; the device gives each strip a lane).
// rec: dm_frame's floats (Ry at [10..18], tx [19], tz [20], offsets [21], [22]).
__host__ inline void frame_geometry(const Cfg& c, const float* rec, FrameGeom& g) {
  const Pose p = pose_of(c, rec[10], rec[12], rec[16], rec[18], rec[19], rec[20], rec[21], rec[22]);
  g.ok = p.ok; g.inside = 0;
  int ux0 = 32767, ux1 = 0, uz0 = 32767, uz1 = 0;
  for (int s = 0; s < kMaxStrips; ++s) {
    bool in = false;
    strip_geometry(c, p, c.cxl[s], c.czl[s], c.tmin[s], c.tmax[s], s < c.P && c.live[s], g.win[s], g.L[s], g.R[s], in);
    g.inside |= in ? 1 << s : 0;
    const Win16 w = g.win[s];
    if (w.w > 0) {
      if (w.x0 < ux0) ux0 = w.x0;
      if (w.x0 + w.w > ux1) ux1 = w.x0 + w.w;
      if (w.z0 < uz0) uz0 = w.z0;
      if (w.z0 + w.h > uz1) uz1 = w.z0 + w.h;
    }
  }
  g.U = ux1 > ux0 ? Win16{(short)ux0, (short)uz0, (short)(ux1 - ux0), (short)(uz1 - uz0)} : Win16{0, 0, 0, 0};
}

// Cover of a strip on map row z: [lo, hi) in cells, multiples of 4, inside the strip's window;
// packed lo | hi << 16 (0 = empty).
__host__ __device__ inline uint32_t row_cover(const Win16& w, const Line& L, const Line& R, int z, int mw) {
  // (selects, no branches: the device runs this right in front of its pixel loop)
  const int wx1 = w.x0 + w.w;
  bool live = (w.w > 0) & (z >= w.z0) & (z < w.z0 + w.h);
  float lo = (float)w.x0, hi = (float)wx1;                  // [lo, hi)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const Line& l = i ? R : L;
    const float rest = l.nz * (float)z + l.k;               // f = nx * x + rest >= 0
    const float xb = -rest * l.inv_nx;
    const float bl = ceilf(xb - 0.25f);                     // nx > 0: x >= -rest / nx
    const float bh = floorf(xb + 0.25f) + 1.0f;             // nx < 0: x <= -rest / nx
    lo = ((l.nx > 0.0f) & (bl > lo)) ? bl : lo;
    hi = ((l.nx < 0.0f) & (bh < hi)) ? bh : hi;
    live = live & !((l.nx == 0.0f) & (rest < 0.0f));
  }
  live = live & (lo < hi);
  lo = lo < (float)wx1 ? lo : (float)wx1;                   // (keeps the conversions in range)
  hi = hi > (float)w.x0 ? hi : (float)w.x0;
  // Whole 128-byte lines of the map (32 cells): the span may reach past the window, where
  // nothing can land (the flush writes the fill value there).  Spans that end inside a line
  // make every kernel that shares the line write a part of it.
  const int ilo = ((int)lo) & ~(kSpanAlign - 1);
  int ihi = ((int)hi + kSpanAlign - 1) & ~(kSpanAlign - 1);
  ihi = ihi > mw ? mw : ihi;
  return live ? (uint32_t)ilo | ((uint32_t)ihi << 16) : 0u;
}

// Cuts the span [lo, hi) of one strip's cover down to a part the cover q does not intersect
// (one step of the owned-span computation; the larger piece is kept when q lies strictly inside).
__host__ __device__ inline void cut_span(int& lo, int& hi, uint32_t q) {
  const int ql = (int)(q & 0xffffu), qh = (int)(q >> 16);
  const bool hit = (qh > lo) & (ql < hi);                   // (an empty cover has qh = 0)
  const bool left = ql <= lo, right = qh >= hi;             // q reaches over this end of the span
  const bool keep_left = (ql - lo) >= (hi - qh);            // q strictly inside: keep the larger piece
  const int nlo = (left & right) ? hi : left ? qh : right ? lo : keep_left ? lo : qh;
  const int nhi = (left & right) ? hi : left ? hi : right ? ql : keep_left ? ql : hi;
  lo = hit ? nlo : lo;
  hi = hit ? nhi : hi;
}

// The owned span of strip p on a row: its cover cut by the other strips' covers, taken in the
// order p ^ 1, p ^ 2, ... (the order in which the device's lanes see them).  P2 = 4 or 8.
__host__ __device__ inline uint32_t row_owned(const uint32_t* cover, int p, int P, int P2) {
  int lo = (int)(cover[p] & 0xffffu), hi = (int)(cover[p] >> 16);
  for (int m = 1; m < P2; ++m) {
    const int q = p ^ m;
    if (q < P) cut_span(lo, hi, cover[q]);
  }
  return hi > lo ? (uint32_t)lo | ((uint32_t)hi << 16) : 0u;
}

// Cells x' < x of a row that lie in TWO OR MORE of (up to) four covers -- the frame-wide numbering of the
// shared float4 groups (compact planes) -- and, for the hole test, the cells x' < x in ANY cover:
// inclusion-exclusion over the covers' intersections (a cell in k covers counts C(k,2) - 2 C(k,3) + 3 C(k,4)
// = 1 for k = 2, 3, 4 among the shared ones, k - C(k,2) + C(k,3) - C(k,4) = 1 for k >= 1 in the union).
// An empty cover (0) meets nothing.
__host__ __device__ inline int cover_measures(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int x, int* covered) {
  const int l0 = (int)(c0 & 0xffffu), l1 = (int)(c1 & 0xffffu), l2 = (int)(c2 & 0xffffu), l3 = (int)(c3 & 0xffffu);
  int h0 = (int)(c0 >> 16), h1 = (int)(c1 >> 16), h2 = (int)(c2 >> 16), h3 = (int)(c3 >> 16);
  h0 = h0 < x ? h0 : x; h1 = h1 < x ? h1 : x; h2 = h2 < x ? h2 : x; h3 = h3 < x ? h3 : x;
  auto mx = [](int a, int b) { return a > b ? a : b; };
  auto mn = [](int a, int b) { return a < b ? a : b; };
  auto len = [](int l, int h) { return h > l ? h - l : 0; };
  const int l01 = mx(l0, l1), h01 = mn(h0, h1), l23 = mx(l2, l3), h23 = mn(h2, h3);
  const int pairs = len(l01, h01) + len(mx(l0, l2), mn(h0, h2)) + len(mx(l0, l3), mn(h0, h3)) +
                    len(mx(l1, l2), mn(h1, h2)) + len(mx(l1, l3), mn(h1, h3)) + len(l23, h23);
  const int triples = len(mx(l01, l2), mn(h01, h2)) + len(mx(l01, l3), mn(h01, h3)) +
                      len(mx(l0, l23), mn(h0, h23)) + len(mx(l1, l23), mn(h1, h23));
  const int quad = len(mx(l01, l23), mn(h01, h23));
  if (covered) *covered = len(l0, h0) + len(l1, h1) + len(l2, h2) + len(l3, h3) - pairs + triples - quad;
  return pairs - 2 * triples + 3 * quad;
}
__host__ __device__ inline int shared_before(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, int x) {
  return cover_measures(c0, c1, c2, c3, x, nullptr);
}

__host__ __device__ inline bool in_span(uint32_t span, int x) {
  return x >= (int)(span & 0xffffu) && x < (int)(span >> 16);
}

}  // namespace strip
}  // namespace dm
