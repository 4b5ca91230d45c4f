// Geometry of the LDS-windowed path (host side and the small structs both sides share):
// how a frame is split into parts, which map window a part can reach, the per-launch tables.
// Included by dm_window.hip only.
#pragma once

#include <math.h>
#include <stddef.h>
#include <stdint.h>

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
  int geo;             // band edges: 0 equal steps, 1 geometric towards 0 (band_edge)
};

// Window of one part in map cells; w == 0: the part cannot hit the map.
struct Window {
  int x0, z0, w, h;
};

// Tables of the scatter kernel, one per chunk of frames (= launch): staged into the workspace
// by ONE hipMemcpyAsync per call (dm_window.hip).  Every access is a scalar load of
// wave-uniform data.  (Passing them as kernel
// arguments instead costs as much at the head of the kernel, and the runtime's
// kernel-argument pool then stalls the host every few dozen launches.)
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
//   geo = 0: equal steps (a truncated range of a few metres).
//   geo = 1: a range that runs to where rays leave the map (bound_depth_range: tens of metres,
//            possibly from -D to +D): band edges in geometric progression towards 0 -- the map
//            is dense near the camera, and the far bands are mostly outside it.  Negative depths
//            get 3 of 8 (1 of 4, 1 of 2) bands.
__host__ __device__ inline float band_edge(float dmin, float dmax, int pd, int j, int geo) {
  if (j <= 0) return dmin;
  if (j >= pd) return dmax;
  if (!geo) {
    const float step = (dmax - dmin) / (float)pd;
    return __builtin_fmaf((float)j, step, dmin);
  }
  const int nneg = dmin < 0.0f && dmax > 0.0f ? (pd >= 8 ? 3 : 1) : (dmax <= 0.0f ? pd : 0);
  float e;
  if (j < nneg) {                              // dmin * 0.3^j
    e = dmin;
    for (int i = 0; i < j; ++i) e *= 0.3f;
  } else if (j == nneg && nneg > 0 && nneg < pd) {
    e = 0.0f;
  } else {                                     // dmax * 0.55^(pd - j)
    e = dmax;
    for (int i = j; i < pd; ++i) e *= 0.55f;
  }
  e = e < dmin ? dmin : e;
  return e > dmax ? dmax : e;
}
__host__ __device__ inline void band_bounds(float dmin, float dmax, int pd, int k, float& lo,
                                            float& hi, int geo = 0) {
  lo = band_edge(dmin, dmax, pd, k, geo);
  hi = band_edge(dmin, dmax, pd, k + 1, geo);
}

// Geometric band edges where the depth range reaches beyond the map (or behind the camera).
__host__ inline int band_mode(const dm_params& p) {
  if (!p.has_dmin || !p.has_dmax) return 0;
  const double cells = ((double)p.dmax - (double)p.dmin) / (double)p.res;
  return (p.dmin < 0.0f || cells > (double)(p.mw > p.mh ? p.mw : p.mh)) ? 1 : 0;
}

__host__ Parts choose_parts(const dm_params& p, int min_parts = 1, int pd = 1) {
  // The split of the image into pc column strips x pr row blocks (x pd depth bands) that the
  // measured cost model likes best, among those of at least min_parts image parts (more,
  // narrower parts until the windows fit in LDS).  Model (MI355X, 1024-thread workgroups, one
  // per CU): workgroups run in waves of 256, a workgroup takes ~6 us + 0.3 us per pixel and
  // thread, the merge ~0.37 us per part.  So: fill the 256 CUs, but in whole waves (16 frames
  // of 20 parts are two waves of which the second is a quarter full: 96 us; 16 parts: one),
  // and do not split a small batch further than pays (B = 1, 320x240: 80 parts 43 us, 10
  // parts 15 us).  Strips are multiples of 32 columns (128-byte lines) when W allows it and
  // equal (pc divides the unit count); at most 16 strips x 8 row blocks of >= 16 rows.
  const long frames = (long)p.B * (p.vc ? p.vc : p.dc);
  const int unit = (p.W % 32 == 0) ? 32 : 4;
  const int units = (p.W + unit - 1) / unit;
  const int max_pr = p.H / 16 > 8 ? 8 : (p.H / 16 > 0 ? p.H / 16 : 1);
  Parts best;
  best.pc = 1; best.pr = 1; best.pd = pd; best.geo = pd > 1 ? band_mode(p) : 0; best.wp = units * unit; best.hp = p.H;
  double best_cost = INFINITY;
  int best_n = 0;
  for (int pc = 1; pc <= units && pc <= 16; ++pc) {
    if (units % pc != 0) continue;
    for (int pr = 1; pr <= max_pr; ++pr) {
      const int n = pc * pr;
      const int wp = (units / pc) * unit, hp = (p.H + pr - 1) / pr;
      const double waves = ceil((double)frames * n * pd / 256.0);
      double cost = waves * (6.0 + 0.3 * (double)wp * hp / 1024.0) + 0.37 * n * pd;
      // not enough parts for the caller: only as the last resort (the most parts there are)
      if (n < min_parts) cost = 1e30 - n;
      const bool better = cost < best_cost - 1e-9 ||
                          (fabs(cost - best_cost) <= 1e-9 && (n < best_n || (n == best_n && pc > best.pc)));
      if (better) {
        best_cost = cost; best_n = n;
        best.pc = pc; best.pr = pr; best.wp = wp; best.hp = hp;
      }
    }
  }
  return best;
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
  double mag;          // largest float32 intermediate of the device's pixel arithmetic, in cells
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
  a.mag = fmax(fmax(fabs(t2[0] * inv), fabs(t2[2] * inv)), fmax(fabs((double)f.width_offset), fabs((double)f.height_offset)));
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

// A part's reach in the map can be bounded when no depth is negative (a negative depth projects
// behind the camera: a double cone): trunc_depth_min >= 0 is required, trunc_depth_max is not --
// without it the part's rays are followed until they have left the map.
__host__ bool frustum_bounded(const dm_params& p) {
  if (!p.has_dmin) return false;
  // both ends finite: any sign (the corners of a band are affine in the depth, so the bounding
  // box of its eight corners holds the whole band, through the camera's cell if it spans 0)
  if (p.has_dmax && isfinite(p.dmin) && isfinite(p.dmax)) return p.dmax >= p.dmin;
  if (!(p.dmin >= 0.0f)) return false;          // open far end: forward rays only
  return !p.has_dmax || p.dmax >= p.dmin;       // (dmax may be +inf)
}

// Missing depth truncations (the reference's default is None for both, maps.py:1267-1268): a
// pixel whose depth is so large -- or so negative -- that its ray has left the map cannot land
// in it, so the call may be given finite bounds without changing a single cell.  With the
// axis-aligned rotations a point at depth d lies |d| * sqrt(ax^2 + g^2) metres from the camera in
// the map plane (g = Rp[5] * ay + Rp[8]); the map reaches at most r_far metres from the camera.
// Fills in dmin / dmax (rounded up on a 1.25^k grid so that the per-shape caches keep their
// keys) and returns true; false when no bound exists (a ray that points straight down or up
// never leaves its cell) or the rotations are not the projector's.
__host__ inline bool bound_depth_range(dm_params& p, const dm_frame* f) {
  if (p.has_dmin && p.has_dmax) return false;
  if (!(p.res > 0.0f) || !(p.fx > 0.0f) || !(p.fy > 0.0f)) return false;
  // smallest planar speed of a ray: sqrt(min ax^2 + min g^2) over the image
  const double ax0 = (0.0 - (double)p.cx) / (double)p.fx, ax1 = ((double)(p.W - 1) - (double)p.cx) / (double)p.fx;
  const double ax_min = (ax0 <= 0.0 && ax1 >= 0.0) ? 0.0 : fmin(fabs(ax0), fabs(ax1));
  double ay[2];
  for (int i = 0; i < 2; ++i) {
    double yr = i ? (double)(p.H - 1) : 0.0;
    if (p.flip_h) yr = (double)(p.H - 1) - yr;
    ay[i] = (yr - (double)p.cy) / (double)p.fy;
  }
  double worst = 0.0;
  for (int b = 0; b < p.B; ++b) {
    const float* rp = f[b].Rp;
    const float* ry = f[b].Ry;
    if (!(rp[0] == 1.0f && rp[1] == 0.0f && rp[2] == 0.0f && rp[3] == 0.0f && rp[6] == 0.0f)) return false;
    if (p.to_global && !(ry[1] == 0.0f && ry[3] == 0.0f && ry[4] == 1.0f && ry[5] == 0.0f && ry[7] == 0.0f))
      return false;
    const double g0 = (double)rp[5] * ay[0] + (double)rp[8], g1 = (double)rp[5] * ay[1] + (double)rp[8];
    const double g_min = (g0 > 0.0) == (g1 > 0.0) ? fmin(fabs(g0), fabs(g1)) : 0.0;
    const double speed = sqrt(ax_min * ax_min + g_min * g_min);
    if (!(speed > 1e-3)) return false;
    // the camera's cell and its distance to the farthest map corner, in cells (+ slack)
    const double inv = 1.0 / (double)p.res;
    const double xd = (p.to_global ? (double)f[b].tx * inv : 0.0) + (double)f[b].width_offset;
    double zd = (p.to_global ? (double)f[b].tz * inv : 0.0) + (double)f[b].height_offset;
    if (p.flip_h) zd = (double)(p.mh - 1) - zd;
    const double dx = fmax(fabs(xd), fabs(xd - (double)p.mw)), dz = fmax(fabs(zd), fabs(zd - (double)p.mh));
    const double r_far = (sqrt(dx * dx + dz * dz) + 4.0) * (double)p.res;
    const double d = r_far / speed;
    if (!(d == d) || !isfinite(d)) return false;
    if (d > worst) worst = d;
  }
  double grid = 1.0;                              // next 1.25^k above 1.01 * worst
  while (grid < 1.01 * worst) grid *= 1.25;
  if (grid > 1e6) return false;
  if (!p.has_dmax) { p.has_dmax = 1; p.dmax = (float)grid; }
  if (!p.has_dmin) { p.has_dmin = 1; p.dmin = -(float)grid; }
  return p.dmax >= p.dmin;
}

__host__ Window part_window(const dm_params& p, const FrameAffine& fa, const PartSlopes& ps,
                            bool bounded, float dlo, float dhi) {
  if (ps.empty) return Window{0, 0, 0, 0};
  // The 2 cells of slack below cover the float32 rounding of the device's cell coordinates
  // (a few ulp of the largest intermediate, x / res before the offset is added) while that
  // intermediate stays below 2^20 cells (ulp 1/8 cell).  Farther from the origin -- a pose
  // millions of cells out, cancelled by the offsets -- no window is claimed: the whole map,
  // which does not fit in LDS, so the call takes the global-atomic path.
  if (!bounded || !fa.finite || !(fa.mag < 1048576.0)) return Window{0, 0, p.mw, p.mh};
  double lo_x = INFINITY, hi_x = -INFINITY, lo_z = INFINITY, hi_z = -INFINITY;
  double poison = 0.0;                         // NaN as soon as one corner is not finite
  const bool open_far = !p.has_dmax || !isfinite(dhi);
  // Open far end: every pixel's cells lie on a ray from the camera's cell (xd, zd).  Inside the
  // map such a cell is at most r_map away from it; the corner rays are followed to 2 r_map, which
  // puts the chord between two of them beyond r_map as long as they span less than 120
  // degrees (checked below: else the whole map).
  double r_far = 0.0;
  if (open_far) {
    const double cx[2] = {0.0, (double)p.mw}, cz[2] = {0.0, (double)p.mh};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j) {
        const double d = hypot(cx[i] - fa.xd, cz[j] - fa.zd);
        if (d > r_far) r_far = d;
      }
    r_far = 2.0 * r_far + 4.0;
  }
  double dirx[4], dirz[4];
  for (int qi = 0; qi < 2; ++qi)
    for (int ri = 0; ri < 2; ++ri) {
      const double sx = fa.xa * ps.ax[qi] + fa.xb * ps.ay[ri] + fa.xc;
      const double sz = fa.za * ps.ax[qi] + fa.zb * ps.ay[ri] + fa.zc;
      dirx[qi * 2 + ri] = sx; dirz[qi * 2 + ri] = sz;
      double far = dhi;
      if (open_far) {
        const double len = hypot(sx, sz);
        if (!(len > 1e-9)) return Window{0, 0, p.mw, p.mh};      // a ray straight down / up
        far = r_far / len;
        if (far < dlo) far = dlo;
      }
      const double zs[2] = {dlo, far};
      for (int zi = 0; zi < 2; ++zi) {
        const double xf = zs[zi] * sx + fa.xd, zf = zs[zi] * sz + fa.zd;
        lo_x = xf < lo_x ? xf : lo_x; hi_x = xf > hi_x ? xf : hi_x;
        lo_z = zf < lo_z ? zf : lo_z; hi_z = zf > hi_z ? zf : hi_z;
        poison += xf * 0.0 + zf * 0.0;
      }
    }
  if (open_far) {
    for (int i = 0; i < 4; ++i)
      for (int j = i + 1; j < 4; ++j) {
        const double dot = dirx[i] * dirx[j] + dirz[i] * dirz[j];
        const double norm = hypot(dirx[i], dirz[i]) * hypot(dirx[j], dirz[j]);
        if (!(dot > -0.5 * norm)) return Window{0, 0, p.mw, p.mh};   // >= 120 degrees apart
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

}  // namespace
}  // namespace dm
