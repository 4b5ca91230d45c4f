// Geometry of the strip path (dm_strip.hip): which map cells each column strip of a frame can
// reach, evaluated by the SAME code on the device (inside k_strip_scatter / k_strip_merge) and
// on the host (sizing, eligibility, the CPU tests through dm_debug_strip_geometry).
//
// A frame is cut into P column strips (pr = pd = 1).  With the projector's axis-aligned
// rotations (pitch about X, yaw about Y) the map position of a pixel with ray slopes (ax, ay)
// and depth d is
//     c = A + d * (ax * U + g(ay) * C),     g(ay) = 1 + kappa * ay,
// A = the camera's cell, U / C = the camera's right / forward direction in cells per metre.
// So every pixel of a strip lies in the CONE  { A + b * (t * U + C) : b >= 0, t in [tmin, tmax] }
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
// Double precision, no libm calls (IEEE +,-,*,/ and floor/ceil only): identical results on
// host and device.
#pragma once

#include <math.h>
#include <stdint.h>

#include "dm_window_geometry.hpp"

namespace dm {
namespace strip {

constexpr int kMaxStrips = 8;

// Pose-independent part of the geometry: one per call, a kernel argument.
struct Cfg {
  int P;                        // column strips
  int mw, mh;
  int flip_h, to_global;
  double ax_lo[kMaxStrips];     // (q - cx) / fx of the strip's first and last live column
  double ax_hi[kMaxStrips];
  int live[kMaxStrips];         // 0: nothing left of the strip after clip_border
  double ay_lo, ay_hi;          // min / max of (y - cy) / fy over the live rows
  double dmin, dmax;            // depth range, 0 <= dmin <= dmax < inf
  double res_inv;
};

// One frame's geometry (L1): windows and cone edges of every strip.
// Edge lines in cell coordinates: a cell centre (x, z) is inside the widened cone of strip p
// iff  L.nx * x + L.nz * z + L.k >= 0  and  R.nx * x + R.nz * z + R.k >= 0.
struct Line { double nx, nz, k, inv_nx; };
struct FrameGeom {
  Win16 win[kMaxStrips];
  Win16 U;                      // bounding box of the windows, x aligned to 4
  Line L[kMaxStrips], R[kMaxStrips];
  int ok;                       // 0: the cone model does not apply to this frame
};

__host__ __device__ inline double dabs(double v) { return v < 0.0 ? -v : v; }
__host__ __device__ inline double dmin2(double a, double b) { return a < b ? a : b; }
__host__ __device__ inline double dmax2(double a, double b) { return a > b ? a : b; }

// Frame record (dm_frame's first 23 floats) -> affine coefficients, as frame_affine() in
// dm_window_geometry.hpp:  xf = d * (xa*ax + xb*ay + xc) + xd,  zf = d * (za*ax + zb*ay + zc) + zd.
struct Affine { double xa, xb, xc, xd, za, zb, zc, zd, mag; };

__host__ __device__ inline Affine frame_affine_f(const Cfg& c, const float* f) {
  // f: Rp[0..8], cam_h [9], Ry[10..18], tx [19], tz [20], wo [21], ho [22]
  double L[3][3], G[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) L[i][j] = (double)f[3 * j + i];
  const double t1[3] = {0.0, (double)f[9], 0.0};
  double t2[3];
  if (c.to_global) {
    double Y[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Y[i][j] = (double)f[10 + 3 * j + i];
    const double tr[3] = {(double)f[19], 0.0, (double)f[20]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) G[i][j] = Y[i][0] * L[0][j] + Y[i][1] * L[1][j] + Y[i][2] * L[2][j];
      t2[i] = Y[i][0] * t1[0] + Y[i][1] * t1[1] + Y[i][2] * t1[2] + tr[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) G[i][j] = L[i][j];
      t2[i] = t1[i];
    }
  }
  Affine a;
  const double inv = c.res_inv;
  a.xa = G[0][0] * inv; a.xb = G[0][1] * inv; a.xc = G[0][2] * inv;
  a.xd = t2[0] * inv + (double)f[21];
  double za = G[2][0] * inv, zb = G[2][1] * inv, zc = G[2][2] * inv;
  double zd = t2[2] * inv + (double)f[22];
  // magnitude of the float32 intermediates the device's pixel arithmetic goes through
  // (x2 / res before the offset is added): its rounding error is a few ulp of this
  a.mag = dmax2(dmax2(dabs(t2[0] * inv), dabs(t2[2] * inv)), dmax2(dabs((double)f[21]), dabs((double)f[22])));
  if (c.flip_h) { za = -za; zb = -zb; zc = -zc; zd = (double)(c.mh - 1) - zd; }
  a.za = za; a.zb = zb; a.zc = zc; a.zd = zd;
  return a;
}

// Cells of slack around everything derived from exact arithmetic: the device computes cell
// coordinates in float32 (error: a few ulp of the largest intermediate, i.e. of
// mag + reach), the reference does the same float32 operations, so the "true" cell is the
// float32 one.  2 cells cover map coordinates up to ~2^17; beyond that the slack grows with
// the magnitude (8 ulp_f32), and past 16 cells the strip path is not used at all.
__host__ __device__ inline double slack_cells(const Affine& a, double reach) {
  const double m = a.mag + reach + dabs(a.xd) + dabs(a.zd);
  return 2.0 + 8.0 * m * (1.0 / 8388608.0);
}

__host__ __device__ inline bool finite_d(double v) { return v == v && v - v == 0.0; }

// Corner k (0..7) of strip p's truncated cone: bit 0 = near/far, bit 1 = ax lo/hi, bit 2 = ay lo/hi.
__host__ __device__ inline void cone_corner(const Cfg& c, const Affine& a, double ax_lo, double ax_hi,
                                            int k, double& xf, double& zf) {
  const double d = (k & 1) ? c.dmax : c.dmin;
  const double ax = (k & 2) ? ax_hi : ax_lo;
  const double ay = (k & 4) ? c.ay_hi : c.ay_lo;
  xf = d * (a.xa * ax + a.xb * ay + a.xc) + a.xd;
  zf = d * (a.za * ax + a.zb * ay + a.zc) + a.zd;
}

// Window from the bounding box [lx, hx] x [lz, hz] of the eight corners.
__host__ __device__ inline Win16 window_of(const Cfg& c, double lx, double hx, double lz, double hz,
                                           double slack) {
  double x0 = floor(lx + 0.5) - slack, x1 = floor(hx + 0.5) + slack + 1.0;
  double z0 = floor(lz + 0.5) - slack, z1 = floor(hz + 0.5) + slack + 1.0;
  x0 = floor(x0); z0 = floor(z0); x1 = ceil(x1); z1 = ceil(z1);
  if (x0 < 0.0) x0 = 0.0;
  if (z0 < 0.0) z0 = 0.0;
  if (x1 > (double)c.mw) x1 = (double)c.mw;
  if (z1 > (double)c.mh) z1 = (double)c.mh;
  if (!(x0 < x1) || !(z0 < z1)) return Win16{0, 0, 0, 0};
  const int ix0 = ((int)x0) & ~3;
  int ixe = ((int)x1 + 3) & ~3;                 // mw % 4 == 0 is a precondition
  if (ixe > c.mw) ixe = c.mw;
  return Win16{(short)ix0, (short)(int)z0, (short)(ixe - ix0), (short)((int)z1 - (int)z0)};
}

// The cone's edge parameters of strip p: t = ax / g over the four corner rays.  Returns false
// when some row of the image does not look forward (g <= 0): no cone then.
struct ConeBasis { double Ux, Uz, Cx, Cz, kappa, sgn; bool ok; };

__host__ __device__ inline ConeBasis cone_basis(const Cfg& c, const Affine& a) {
  ConeBasis b;
  b.Ux = a.xa; b.Uz = a.za; b.Cx = a.xc; b.Cz = a.zc;
  const double cc = b.Cx * b.Cx + b.Cz * b.Cz;
  const double vv = a.xb * a.xb + a.zb * a.zb;
  const double uu = b.Ux * b.Ux + b.Uz * b.Uz;
  // V = (xb, zb) must be parallel to C (true for pitch about X + yaw about Y): g = 1 + kappa * ay
  const double cross_vc = a.xb * b.Cz - a.zb * b.Cx;
  b.kappa = cc > 0.0 ? (a.xb * b.Cx + a.zb * b.Cz) / cc : 0.0;
  const double sigma = b.Cx * b.Uz - b.Cz * b.Ux;           // cr(C, U)
  b.sgn = sigma < 0.0 ? -1.0 : 1.0;
  const double g0 = 1.0 + b.kappa * c.ay_lo, g1 = 1.0 + b.kappa * c.ay_hi;
  b.ok = cc > 0.0 && uu > 0.0 && finite_d(cc) && finite_d(uu) && finite_d(vv) &&
         cross_vc * cross_vc <= 1e-18 * vv * cc && dabs(sigma) * dabs(sigma) > 1e-6 * cc * uu &&
         g0 > 1e-3 && g1 > 1e-3;
  return b;
}

__host__ __device__ inline void cone_t_range(const Cfg& c, const ConeBasis& b, int p, double& tmin,
                                             double& tmax) {
  const double g0 = 1.0 + b.kappa * c.ay_lo, g1 = 1.0 + b.kappa * c.ay_hi;
  const double t[4] = {c.ax_lo[p] / g0, c.ax_lo[p] / g1, c.ax_hi[p] / g0, c.ax_hi[p] / g1};
  tmin = dmin2(dmin2(t[0], t[1]), dmin2(t[2], t[3]));
  tmax = dmax2(dmax2(t[0], t[1]), dmax2(t[2], t[3]));
}

// Edge line of the cone through A with direction D = t * U + C.  left: the edge at tmin
// (inside: t >= tmin), else the edge at tmax.  The margin (half a cell plus the slack, in
// the line's own units) is folded into k.
__host__ __device__ inline Line cone_edge(const ConeBasis& b, const Affine& a, double t, bool left,
                                          double slack) {
  const double Dx = t * b.Ux + b.Cx, Dz = t * b.Uz + b.Cz;
  Line l;
  if (left) { l.nx = -b.sgn * Dz; l.nz = b.sgn * Dx; }      // sgn * cr(D, w)
  else      { l.nx = b.sgn * Dz;  l.nz = -b.sgn * Dx; }     // sgn * cr(w, D)
  l.k = -l.nx * a.xd - l.nz * a.zd + (0.5 + slack) * (dabs(l.nx) + dabs(l.nz));
  l.inv_nx = l.nx != 0.0 ? 1.0 / l.nx : 0.0;
  return l;
}

// The whole L1 geometry of one frame, serially (host; the device spreads the same calls over
// the lanes of a wave).
__host__ __device__ inline void frame_geometry(const Cfg& c, const float* frame_rec, FrameGeom& g) {
  const Affine a = frame_affine_f(c, frame_rec);
  const ConeBasis b = cone_basis(c, a);
  bool fin = finite_d(a.xa) && finite_d(a.xb) && finite_d(a.xc) && finite_d(a.xd) &&
             finite_d(a.za) && finite_d(a.zb) && finite_d(a.zc) && finite_d(a.zd);
  const double reach = c.dmax * (dabs(a.xa) + dabs(a.xb) + dabs(a.xc) + dabs(a.za) + dabs(a.zb) + dabs(a.zc));
  const double slack = slack_cells(a, reach);
  g.ok = b.ok && fin && slack <= 16.0;
  int ux0 = c.mw, ux1 = 0, uz0 = c.mh, uz1 = 0;
  for (int p = 0; p < kMaxStrips; ++p) {
    g.win[p] = Win16{0, 0, 0, 0};
    g.L[p] = Line{0.0, 0.0, 0.0, 0.0};
    g.R[p] = Line{0.0, 0.0, 0.0, 0.0};
    if (p >= c.P || !c.live[p] || !g.ok) continue;
    double lx = INFINITY, hx = -INFINITY, lz = INFINITY, hz = -INFINITY;
    for (int k = 0; k < 8; ++k) {
      double xf, zf;
      cone_corner(c, a, c.ax_lo[p], c.ax_hi[p], k, xf, zf);
      lx = dmin2(lx, xf); hx = dmax2(hx, xf); lz = dmin2(lz, zf); hz = dmax2(hz, zf);
    }
    const Win16 w = window_of(c, lx, hx, lz, hz, slack);
    g.win[p] = w;
    double tmin, tmax;
    cone_t_range(c, b, p, tmin, tmax);
    g.L[p] = cone_edge(b, a, tmin, true, slack);
    g.R[p] = cone_edge(b, a, tmax, false, slack);
    if (w.w > 0) {
      if (w.x0 < ux0) ux0 = w.x0;
      if (w.x0 + w.w > ux1) ux1 = w.x0 + w.w;
      if (w.z0 < uz0) uz0 = w.z0;
      if (w.z0 + w.h > uz1) uz1 = w.z0 + w.h;
    }
  }
  g.U = ux1 > ux0 ? Win16{(short)ux0, (short)uz0, (short)(ux1 - ux0), (short)(uz1 - uz0)} : Win16{0, 0, 0, 0};
}

// Cover of strip p on map row z: [lo, hi) in cells, multiples of 4, inside the strip's window;
// packed lo | hi << 16 (0 = empty).
__host__ __device__ inline uint32_t row_cover(const Win16& w, const Line& L, const Line& R, int z) {
  if (w.w <= 0 || z < w.z0 || z >= w.z0 + w.h) return 0u;
  double lo = (double)w.x0, hi = (double)(w.x0 + w.w);      // [lo, hi)
  const Line* e[2] = {&L, &R};
  for (int i = 0; i < 2; ++i) {
    const Line& l = *e[i];
    const double rest = l.nz * (double)z + l.k;             // f = nx * x + rest >= 0
    if (l.nx > 0.0) {                                       // x >= -rest / nx
      const double b = ceil(-rest * l.inv_nx - 1e-9);
      if (b > lo) lo = b;
    } else if (l.nx < 0.0) {                                // x <= -rest / nx
      const double b = floor(-rest * l.inv_nx + 1e-9) + 1.0;
      if (b < hi) hi = b;
    } else if (rest < 0.0) {
      return 0u;
    }
  }
  if (!(lo < hi)) return 0u;
  const int ilo = ((int)lo) & ~3;
  int ihi = ((int)hi + 3) & ~3;
  if (ihi > w.x0 + w.w) ihi = w.x0 + w.w;
  return (uint32_t)ilo | ((uint32_t)ihi << 16);
}

__host__ __device__ inline bool in_cover(uint32_t cover, int x) {
  return x >= (int)(cover & 0xffffu) && x < (int)(cover >> 16);
}

}  // namespace strip
}  // namespace dm
