// Per-pixel geometry of the projector: depth pixel -> (map cell, height).
//
// This is the arithmetic contract of the whole library: float32 throughout,
// one rounding per written operation (the translation unit is compiled with
// -ffp-contract=off), true IEEE division, and the two rotations evaluated as
// the 3-term FMA chain  fma(p2, R[6+i], fma(p1, R[3+i], p0 * R[i]))  -- the
// order in which the reference's einsum->bmm accumulates on its CPU path
// (reference dungeon_maps/utils.py:329).  Cell indices therefore match the
// reference bit for bit; see tests/test_hip_parity.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dungeon_maps_amd.h"
#include "../../include/dungeon_maps_amd_debug.h"

namespace dm {

// buffer_store_dwordx4 with the scalar offset in an SGPR (the fill duty's addressing form).  The
// memory pipeline reads a store's data registers for a few cycles AFTER the instruction has
// issued; for stores of more than 8 bytes the compiler's hazard recogniser covers that with wait
// states in front of a vector instruction that overwrites them -- except for buffer stores whose
// scalar-offset field holds a register, which it takes to be free of the hazard.  On gfx950 they
// are not: with the four copies of the fill value reloaded from a spill into registers that the
// very next instruction reused (v_cvt_f32_i32 of an image row), the first dword of the store came
// out as that row number in lanes 12-15 and 28-31 -- reduction='mean' on the window path, found
// by the parity campaign's mean mode (tests/campaigns/parity_campaign.py).  So: the data stays an
// input of two wait states behind the store, which keeps its registers unmodified until then.
#ifdef __HIPCC__
// Cache policy of the fill duties' stores: nt (non-temporal, bit 1 of the cache-policy operand).  The
// fill value is write-once data nobody reads back soon -- most of every output map -- and kept out
// of the caches it stops evicting what IS read again: the pixel lists of the value pass, the cells
// the batch fuse reads.  Measured (bench.py, same box, default policy -> nt): cfg2 step 55.3 ->
// 52.8 us (1.16 -> 1.21 M frames/s), cfg3 launch 1 700 -> 1 578 us, cfg5 (335 MB of fill per
// launch, more than the Infinity Cache holds) 105 -> 83 us; the back-to-back cfg2 launch figure,
// whose 84 MB of output the Infinity Cache otherwise absorbs call after call, 41.5 -> 42.3 us.
// So the strip path's projection kernel comes in both policies (NT_FILL) and the host picks: nt where
// the call's own batch fuse follows (it reads the flushed cells, never the fill) or where the output
// is larger than the Infinity Cache could keep anyway (dm_strip.hip nt_fill_pays); the default
// policy for a plain call whose maps fit the cache, where whoever reads them next finds them there.
// (sc1 alone -- written through -- changes nothing; sc0 / sc1 on top of nt neither.  The mask
// bytes that go with the fill value keep the default policy: four bytes per lane want the L2's write
// combining -- nt on them too: cfg5 81 -> 90 us, cfg3 1 590 -> 1 610 us.)
constexpr int kFillCachePolicy = 2;
template <int POLICY = kFillCachePolicy, class V4>
__device__ inline void buffer_store_b128_at_scalar_offset(V4 data, __amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
  __builtin_amdgcn_raw_buffer_store_b128(data, rsrc, voffset, soffset, POLICY);
  asm volatile("s_nop 1" :: "v"(data));
}
#endif


// Call-wide constants in the form the kernels consume them.
struct View {
  int H, W, mh, mw;
  int clip;            // 0 = off
  int flip_h, to_global;
  int has_dmin, has_dmax, has_hmax;
  float cx, cy, fx, fy, res;
  float dmin, dmax, hmax;
  float Hm1, mhm1;     // float(H-1), float(mh-1)
  float fmw, fmh;      // float(mw), float(mh)
};

__host__ inline View make_view(const dm_params& p) {
  View v;
  v.H = p.H; v.W = p.W; v.mh = p.mh; v.mw = p.mw;
  v.clip = p.clip_border > 0 ? p.clip_border : 0;
  v.flip_h = p.flip_h != 0; v.to_global = p.to_global != 0;
  v.has_dmin = p.has_dmin != 0; v.has_dmax = p.has_dmax != 0; v.has_hmax = p.has_hmax != 0;
  v.cx = p.cx; v.cy = p.cy; v.fx = p.fx; v.fy = p.fy; v.res = p.res;
  v.dmin = p.dmin; v.dmax = p.dmax; v.hmax = p.hmax;
  v.Hm1 = (float)(p.H - 1); v.mhm1 = (float)(p.mh - 1);
  v.fmw = (float)p.mw; v.fmh = (float)p.mh;
  return v;
}

// One frame's camera state pulled into registers (wave-uniform: SGPRs).
struct Cam {
  float p[9];   // pitch rotation
  float y[9];   // yaw rotation
  float h, tx, tz, wo, ho;
};

__device__ inline Cam load_cam(const dm_frame* __restrict__ f) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 9; ++i) { c.p[i] = f->Rp[i]; c.y[i] = f->Ry[i]; }
  c.h = f->cam_height; c.tx = f->tx; c.tz = f->tz;
  c.wo = f->width_offset; c.ho = f->height_offset;
  return c;
}

struct Hit {
  int cell;     // zb * mw + xb, or -1 when the pixel does not land in the map
  float y;      // height in the output frame (y2)
};

// Image column/row -> normalised ray slopes, reference maps.py:670-678.
__device__ inline float ray_x(const View& v, int q) { return ((float)q - v.cx) / v.fx; }
__device__ inline float ray_y(const View& v, int r) {
  float yr = (float)r;
  if (v.flip_h) yr = v.Hm1 - yr;
  return (yr - v.cy) / v.fy;
}

// The projector proper.  `ax`, `ay` are ray_x(q), ray_y(r); `ok` carries the
// validity known so far (valid_map, border clip).
__device__ inline Hit project_pixel(const View& v, const Cam& c, float z, float ax,
                                    float ay, bool ok) {
  // camera space (maps.py:677-678)
  const float X = ax * z;
  const float Y = ay * z;
  // depth truncation (maps.py:537-544); NaN compares false
  if (v.has_dmax) ok = ok && (z <= v.dmax);
  if (v.has_dmin) ok = ok && (z >= v.dmin);
  // pitch about X then lift by the camera height (maps.py:790-797)
  const float x1 = __builtin_fmaf(z, c.p[6], __builtin_fmaf(Y, c.p[3], X * c.p[0])) + 0.0f;
  const float y1 = __builtin_fmaf(z, c.p[7], __builtin_fmaf(Y, c.p[4], X * c.p[1])) + c.h;
  const float z1 = __builtin_fmaf(z, c.p[8], __builtin_fmaf(Y, c.p[5], X * c.p[2])) + 0.0f;
  if (v.has_hmax) ok = ok && (y1 <= v.hmax);       // maps.py:286-288
  float x2 = x1, y2 = y1, z2 = z1;
  if (v.to_global) {                               // maps.py:884-892
    x2 = __builtin_fmaf(z1, c.y[6], __builtin_fmaf(y1, c.y[3], x1 * c.y[0])) + c.tx;
    y2 = __builtin_fmaf(z1, c.y[7], __builtin_fmaf(y1, c.y[4], x1 * c.y[1])) + 0.0f;
    z2 = __builtin_fmaf(z1, c.y[8], __builtin_fmaf(y1, c.y[5], x1 * c.y[2])) + c.tz;
  }
  // quantise, round half up (maps.py:1004-1013)
  float xf = x2 / v.res + c.wo;
  float zf = z2 / v.res + c.ho;
  if (v.flip_h) zf = v.mhm1 - zf;
  xf = __builtin_floorf(xf + 0.5f);
  zf = __builtin_floorf(zf + 0.5f);
  // canvas bounds (maps.py:1150-1158) tested in float: NaN/inf fall out here,
  // exactly like the reference's INT64_MIN after .to(int64)
  ok = ok && (xf >= 0.0f) && (xf < v.fmw) && (zf >= 0.0f) && (zf < v.fmh);
  Hit h;
  h.cell = ok ? (int)zf * v.mw + (int)xf : -1;
  h.y = y2;
  return h;
}

__device__ inline bool border_ok(const View& v, int r, int q) {
  // maps.py:48-70
  return v.clip == 0 ||
         (r >= v.clip && r < v.H - v.clip && q >= v.clip && q < v.W - v.clip);
}

// ---------------------------------------------------------------------------
// Fast-path geometry (dm_window.hip).  Same results as project_pixel, fewer
// instructions:
//  * a/b with y = RN(1/b) precomputed on the host:  q0 = a*y,
//    q = fma(fma(-b, q0, a), y, q0) is the correctly rounded IEEE quotient for
//    1e-25 <= |a| <= 1e25 (Markstein; brute-forced by tools/check_fma_division.c).
//  * the rotations built by rotate([1,0,0],.) / rotate([0,1,0],.) have exact
//    0/1 entries; their zero terms do not change a finite FMA chain (x*0 = +-0,
//    fma(a,b,+-0) = RN(a*b)), and a non-finite chain is non-finite either way
//    and lands outside the map.  The host checks the pattern for every frame
//    of the call (axis_aligned) and otherwise the full chain is used.
struct FastView {
  View v;
  float res_inv, fx_inv, fy_inv;   // RN(1/res), RN(1/fx), RN(1/fy)
};

// Correctly rounded a / b from y = RN(1/b) (3 instructions, no branch).
// Exact for 1e-25 <= |a| <= 1e25.  Outside that range the result can differ
// from IEEE division, but never in a way the projector can see, PROVIDED
// 1e-6 <= b <= 1e6 (checked on the host, else FAST_DIV is off): a huge |a|
// gives a huge or non-finite quotient either way (outside every map); a tiny
// |a| gives a quotient below 1e-19 either way, which vanishes when the offset
// and the 0.5 are added; NaN/inf stay non-finite.
__device__ inline float div_markstein(float a, float b, float y) {
  const float q0 = a * y;
  return __builtin_fmaf(__builtin_fmaf(-b, q0, a), y, q0);
}

template <bool FAST_DIV>
__device__ inline float div_f(float a, float b, float y) {
  return FAST_DIV ? div_markstein(a, b, y) : a / b;
}

template <bool FAST_DIV>
__device__ inline float fray_x(const FastView& fv, int q) {
  return div_f<FAST_DIV>((float)q - fv.v.cx, fv.v.fx, fv.fx_inv);
}
template <bool FAST_DIV>
__device__ inline float fray_y(const FastView& fv, int r) {
  float yr = (float)r;
  if (fv.v.flip_h) yr = fv.v.Hm1 - yr;
  return div_f<FAST_DIV>(yr - fv.v.cy, fv.v.fy, fv.fy_inv);
}

// Floored cell coordinates (as floats) + height of one pixel.  `z` must
// already be NaN for pixels rejected by the depth range / valid map / border
// (NaN propagates to xf, zf); the height truncation is returned in `low`.
template <bool AXIS_ALIGNED, bool FAST_DIV>
__device__ inline void project_point(const FastView& fv, const Cam& c, float z, float ax,
                                     float ay, float& xf, float& zf, float& y2, bool& low) {
  const View& v = fv.v;
  const float X = ax * z;
  const float Y = ay * z;
  float x1, y1, z1;
  if (AXIS_ALIGNED) {
    x1 = X;
    y1 = __builtin_fmaf(z, c.p[7], Y * c.p[4]) + c.h;
    z1 = __builtin_fmaf(z, c.p[8], Y * c.p[5]);
  } else {
    x1 = __builtin_fmaf(z, c.p[6], __builtin_fmaf(Y, c.p[3], X * c.p[0])) + 0.0f;
    y1 = __builtin_fmaf(z, c.p[7], __builtin_fmaf(Y, c.p[4], X * c.p[1])) + c.h;
    z1 = __builtin_fmaf(z, c.p[8], __builtin_fmaf(Y, c.p[5], X * c.p[2])) + 0.0f;
  }
  low = !v.has_hmax || (y1 <= v.hmax);
  float x2 = x1, z2 = z1;
  y2 = y1;
  if (v.to_global) {
    if (AXIS_ALIGNED) {
      x2 = __builtin_fmaf(z1, c.y[6], x1 * c.y[0]) + c.tx;
      z2 = __builtin_fmaf(z1, c.y[8], x1 * c.y[2]) + c.tz;
    } else {
      x2 = __builtin_fmaf(z1, c.y[6], __builtin_fmaf(y1, c.y[3], x1 * c.y[0])) + c.tx;
      y2 = __builtin_fmaf(z1, c.y[7], __builtin_fmaf(y1, c.y[4], x1 * c.y[1])) + 0.0f;
      z2 = __builtin_fmaf(z1, c.y[8], __builtin_fmaf(y1, c.y[5], x1 * c.y[2])) + c.tz;
    }
  }
  xf = div_f<FAST_DIV>(x2, v.res, fv.res_inv) + c.wo;
  zf = div_f<FAST_DIV>(z2, v.res, fv.res_inv) + c.ho;
  if (v.flip_h) zf = v.mhm1 - zf;
  xf = __builtin_floorf(xf + 0.5f);
  zf = __builtin_floorf(zf + 0.5f);
}

// camera_affine_grid of one pixel (maps.py:353-460): where a pixel with depth z and camera-space
// coordinates (X, Y) = (ray_x * z, ray_y * z) lands in the image after the camera moved.  Chain,
// all float32 in the reference's op order: camera_to_local_space (753-800: rp, cam_h) ->
// local_to_global_space(trans_pose) (850-895: ry, tx, tz) -> local_to_camera_space (802-848:
// translate by (0, -cam_h, 0), rotate by -pitch: ri) -> camera_to_image_space (684-751: z_eps =
// z + 1e-7, x / z_eps * fx + cx, flip).  The FULL FMA chains (zero entries included): a non-finite
// depth must give the reference's NaN / inf, which reach the output here.  One definition for the
// stand-alone kernel (dm_points.hip) and the projection kernel that computes the flow from the
// depth it has loaded anyway (dm_window_kernels.hpp, HAS_FLOW).
__device__ inline void flow_pixel(float z, float X, float Y, const float (&rp)[9], float cam_h,
                                  const float (&ry)[9], float tx, float tz, const float (&ri)[9],
                                  float fx, float cx, float fy, float cy, bool flip_h, float Hm1,
                                  float& u, float& w) {
  const float x1 = __builtin_fmaf(z, rp[6], __builtin_fmaf(Y, rp[3], X * rp[0])) + 0.0f;
  const float y1 = __builtin_fmaf(z, rp[7], __builtin_fmaf(Y, rp[4], X * rp[1])) + cam_h;
  const float z1 = __builtin_fmaf(z, rp[8], __builtin_fmaf(Y, rp[5], X * rp[2])) + 0.0f;
  const float x2 = __builtin_fmaf(z1, ry[6], __builtin_fmaf(y1, ry[3], x1 * ry[0])) + tx;
  const float y2 = __builtin_fmaf(z1, ry[7], __builtin_fmaf(y1, ry[4], x1 * ry[1])) + 0.0f;
  const float z2 = __builtin_fmaf(z1, ry[8], __builtin_fmaf(y1, ry[5], x1 * ry[2])) + tz;
  const float x3 = x2 + 0.0f, y3 = y2 + (-cam_h), z3 = z2 + 0.0f;
  const float xc = __builtin_fmaf(z3, ri[6], __builtin_fmaf(y3, ri[3], x3 * ri[0]));
  const float yc = __builtin_fmaf(z3, ri[7], __builtin_fmaf(y3, ri[4], x3 * ri[1]));
  const float zc = __builtin_fmaf(z3, ri[8], __builtin_fmaf(y3, ri[5], x3 * ri[2]));
  const float z_eps = zc + 1e-7f;
  u = xc / z_eps * fx + cx;
  w = yc / z_eps * fy + cy;
  if (flip_h) w = Hm1 - w;
}

// flow_pixel for two pixels at a time: the same float32 operations in the same order on both
// halves of 64-bit register pairs (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 issue once for two
// pixels; the stand-alone kernel is bound by VALU issue -- 85 instructions per pixel one by one,
// profiles/r04_sq_counters.md).  The coefficients come as {c, c} pairs the caller builds once
// (wave-uniform: the compiler keeps them in scalar register pairs).  The two perspective
// divisions stay IEEE divisions per pixel.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct FlowPairs {
  f32x2 rp[9], ry[9], ri[9], cam_h, neg_cam_h, tx, tz, fx, cx, fy, cy, zero, eps, Hm1;
};
__device__ inline void flow_pixel2(f32x2 z, f32x2 X, f32x2 Y, const FlowPairs& c, bool flip_h, f32x2& u, f32x2& w) {
  auto fma2 = [](f32x2 a, f32x2 b, f32x2 d) { return __builtin_elementwise_fma(a, b, d); };
  const f32x2 x1 = fma2(z, c.rp[6], fma2(Y, c.rp[3], X * c.rp[0])) + c.zero;
  const f32x2 y1 = fma2(z, c.rp[7], fma2(Y, c.rp[4], X * c.rp[1])) + c.cam_h;
  const f32x2 z1 = fma2(z, c.rp[8], fma2(Y, c.rp[5], X * c.rp[2])) + c.zero;
  const f32x2 x2 = fma2(z1, c.ry[6], fma2(y1, c.ry[3], x1 * c.ry[0])) + c.tx;
  const f32x2 y2 = fma2(z1, c.ry[7], fma2(y1, c.ry[4], x1 * c.ry[1])) + c.zero;
  const f32x2 z2 = fma2(z1, c.ry[8], fma2(y1, c.ry[5], x1 * c.ry[2])) + c.tz;
  const f32x2 x3 = x2 + c.zero, y3 = y2 + c.neg_cam_h, z3 = z2 + c.zero;
  const f32x2 xc = fma2(z3, c.ri[6], fma2(y3, c.ri[3], x3 * c.ri[0]));
  const f32x2 yc = fma2(z3, c.ri[7], fma2(y3, c.ri[4], x3 * c.ri[1]));
  const f32x2 zc = fma2(z3, c.ri[8], fma2(y3, c.ri[5], x3 * c.ri[2]));
  const f32x2 z_eps = zc + c.eps;
  const f32x2 qu = {xc.x / z_eps.x, xc.y / z_eps.y}, qw = {yc.x / z_eps.x, yc.y / z_eps.y};
  u = qu * c.fx + c.cx;
  w = qw * c.fy + c.cy;
  if (flip_h) w = c.Hm1 - w;
}

// ---------------------------------------------------------------------------
// order-preserving float <-> uint key: k(a) < k(b)  <=>  a < b (with -0 < +0)
__device__ inline uint32_t f2key(float f) {
  uint32_t u = __float_as_uint(f);
  return u ^ ((u & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ inline float key2f(uint32_t k) {
  uint32_t u = k ^ ((k & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu);
  return __uint_as_float(u);
}

// utils.py:489-491 as a function of (map, fill)
__device__ inline uint8_t mask_of(float out, float fill) {
  const float d = out - fill;     // inf - inf = NaN -> unchanged
  return (d != 0.0f && d == d) ? 1 : 0;
}

}  // namespace dm
