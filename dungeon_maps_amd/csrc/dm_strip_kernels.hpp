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

constexpr unsigned kSinks = 256;
constexpr int kGeomBytes = 1024;          // LDS reserved for the FrameGeom in front of the cover table
static_assert(sizeof(strip::FrameGeom) == 592 && sizeof(strip::FrameGeom) <= kGeomBytes, "FrameGeom layout (tests/test_hip_strip.py reads it) and its LDS slot");

struct StripArgs {
  int W, H;
  int clip, flip_h, to_global;
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
  uint32_t* g_covers;         // (B, max_rows, P)
  int* status;                // set non-zero when a frame's geometry does not fit the launch
  float* sink;                // kSinks x 64 bytes: where fill stores with nothing to write go
  strip::Cfg cfg;
};

// 64-lane min / max of doubles over groups of `width` lanes (width = 4 or 8: lanes of one strip)
__device__ inline double wave_min(double v, int width) {
  for (int m = 1; m < width; m <<= 1) { const double o = __shfl_xor(v, m, 64); v = o < v ? o : v; }
  return v;
}
__device__ inline double wave_max(double v, int width) {
  for (int m = 1; m < width; m <<= 1) { const double o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
  return v;
}

// L1 geometry of one frame by ONE wave: lane = strip * 8 + corner.  Same calls as
// strip::frame_geometry (the host's serial version), spread over the lanes.
__device__ inline void strip_geometry_wave(const strip::Cfg& c, const float (&fr)[23], int lane,
                                           strip::FrameGeom* g) {
  using namespace strip;
  const Affine a = frame_affine_f(c, fr);
  const ConeBasis b = cone_basis(c, a);
  const bool fin = finite_d(a.xa) && finite_d(a.xb) && finite_d(a.xc) && finite_d(a.xd) &&
                   finite_d(a.za) && finite_d(a.zb) && finite_d(a.zc) && finite_d(a.zd);
  const double reach = c.dmax * (dabs(a.xa) + dabs(a.xb) + dabs(a.xc) + dabs(a.za) + dabs(a.zb) + dabs(a.zc));
  const double slack = slack_cells(a, reach);
  const int ok = b.ok && fin && slack <= 16.0;
  const int p = lane >> 3, k = lane & 7;
  // this lane's strip: selected by compares (a lane-indexed kernel-argument array would go
  // through scratch memory)
  double ax_lo = 0.0, ax_hi = 0.0;
  int live_p = 0;
#pragma unroll
  for (int s = 0; s < kMaxStrips; ++s) {
    ax_lo = p == s ? c.ax_lo[s] : ax_lo;
    ax_hi = p == s ? c.ax_hi[s] : ax_hi;
    live_p = p == s ? c.live[s] : live_p;
  }
  const bool live = p < c.P && live_p && ok;
  double xf, zf;
  cone_corner(c, a, ax_lo, ax_hi, k, xf, zf);
  const double lx = wave_min(xf, 8), hx = wave_max(xf, 8), lz = wave_min(zf, 8), hz = wave_max(zf, 8);
  Win16 w = window_of(c, lx, hx, lz, hz, slack);
  if (!live) w = Win16{0, 0, 0, 0};
  // t = ax / g over the four corner rays (lanes k = 0..3 of the strip), then the two edges
  const double g0 = 1.0 + b.kappa * c.ay_lo, g1 = 1.0 + b.kappa * c.ay_hi;
  const double t = ((k & 2) ? ax_hi : ax_lo) / ((k & 1) ? g1 : g0);
  const double tmin = wave_min(t, 4), tmax = wave_max(t, 4);
  if (k == 0) {
    g->win[p] = w;
    g->L[p] = live ? cone_edge(b, a, tmin, true, slack) : Line{0.0, 0.0, 0.0, 0.0};
  }
  if (k == 1) g->R[p] = live ? cone_edge(b, a, tmax, false, slack) : Line{0.0, 0.0, 0.0, 0.0};
  // union window: over the strips (lanes 0, 8, 16, ...)
  int ux0 = w.w > 0 ? w.x0 : 32767, ux1 = w.w > 0 ? w.x0 + w.w : 0;
  int uz0 = w.w > 0 ? w.z0 : 32767, uz1 = w.w > 0 ? w.z0 + w.h : 0;
  for (int m = 8; m < 64; m <<= 1) {
    ux0 = min(ux0, __shfl_xor(ux0, m, 64)); ux1 = max(ux1, __shfl_xor(ux1, m, 64));
    uz0 = min(uz0, __shfl_xor(uz0, m, 64)); uz1 = max(uz1, __shfl_xor(uz1, m, 64));
  }
  if (lane == 0) {
    g->U = ux1 > ux0 ? Win16{(short)ux0, (short)uz0, (short)(ux1 - ux0), (short)(uz1 - uz0)}
                     : Win16{0, 0, 0, 0};
    g->ok = ok;
  }
}

// RED: kMin / kMax.  Always the fast geometry (axis-aligned rotations, exact FMA division),
// 16-byte depth loads.  HAS_VALUE / HAS_VALID / LEAN as in k_window_scatter.
template <int RED, bool HAS_VALID, bool HAS_VALUE, bool LEAN>
__global__ void __launch_bounds__(kScatterThreads)
k_strip_scatter(StripArgs a) {
  constexpr int VEC = 4;
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

  // the frame record: one batch of scalar loads, pinned
  float fr[23];
  {
    const float* tf = a.frames + (size_t)b * 32;
#pragma unroll
    for (int i = 0; i < 23; ++i) fr[i] = tf[i];
    asm volatile("" : "+s"(fr[0]), "+s"(fr[1]), "+s"(fr[2]), "+s"(fr[3]), "+s"(fr[4]), "+s"(fr[5]),
                      "+s"(fr[6]), "+s"(fr[7]), "+s"(fr[8]), "+s"(fr[9]), "+s"(fr[10]), "+s"(fr[11]),
                      "+s"(fr[12]), "+s"(fr[13]), "+s"(fr[14]), "+s"(fr[15]), "+s"(fr[16]),
                      "+s"(fr[17]), "+s"(fr[18]), "+s"(fr[19]), "+s"(fr[20]), "+s"(fr[21]),
                      "+s"(fr[22]));
  }
  strip::FrameGeom* geom = reinterpret_cast<strip::FrameGeom*>(lds + a.table_off);
  uint32_t* covers = reinterpret_cast<uint32_t*>(lds + a.table_off + kGeomBytes / 4);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const float lds_init = a.fill;
  // Wave 0 derives the frame's geometry while the others initialise the whole window region
  // of LDS (the window's size is not known before the geometry is).
  if (wave == 0) {
    strip_geometry_wave(a.cfg, fr, (int)threadIdx.x & 63, geom);
  }
  for (int i = threadIdx.x * 4; i < a.slab_stride + 64; i += kScatterThreads * 4)
    *reinterpret_cast<float4*>(lds + i) = make_float4(lds_init, lds_init, lds_init, lds_init);
  lds_barrier();
  Window w = widen(geom->win[part]);
  Window U = widen(geom->U);
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
  {
    const int wv = wave;
    if (wv >= 12) __builtin_amdgcn_s_setprio(3);
    else if (wv >= 8) __builtin_amdgcn_s_setprio(2);
    else if (wv >= 4) __builtin_amdgcn_s_setprio(1);
  }
  // cover table of the frame: one map row of the union window per thread, every strip's cover
  const bool publisher = part == 0 && chl == 0;
  for (int r = threadIdx.x; r < U.h; r += kScatterThreads) {
    for (int p = 0; p < nparts; ++p) {
      const uint32_t cv = strip::row_cover(geom->win[p], geom->L[p], geom->R[p], U.z0 + r);
      covers[r * nparts + p] = cv;
      if (publisher) a.g_covers[((size_t)b * a.max_rows + r) * nparts + p] = cv;
    }
  }

  // Fill duty (as in k_window_scatter): map rows part, part + P, ... outside the union window
  // (a store that has nothing to write -- inside U, or past the end -- goes to this workgroup's
  // sink in the workspace: unlike k_window_scatter's merge, nothing rewrites all of U later, and
  // other workgroups write owned groups of U while this one fills)
  const int g4 = a.mw >> 2;
  const int fill_rows = (a.mh - part + nparts - 1) / nparts;
  const size_t map_base = ((size_t)b * a.oc_total + ch) * (size_t)a.mh * a.mw;
  float* const sink = a.sink + (((unsigned)part + (unsigned)nparts * ((unsigned)chl + (unsigned)bl)) % kSinks) * 16u;
  const bool do_fill = a.out != nullptr && fill_rows > 0;
  const int fill_total = do_fill ? fill_rows * g4 : 0;
  const int fill_steps = (fill_total + kScatterThreads - 1) / kScatterThreads;
  const float g4_inv = 1.0f / (float)g4;
  int fs = 0;
  auto fill_step = [&]() {
    const int i = fs * kScatterThreads + (int)threadIdx.x;
    ++fs;
    int k = (int)((float)i * g4_inv);
    k -= (k * g4 > i);
    k += ((k + 1) * g4 <= i);
    const int g = i - k * g4;
    const int r = part + k * nparts, x = g << 2;
    const bool skip = i >= fill_total || ((unsigned)(r - U.z0) < (unsigned)U.h &&
                                          (unsigned)(x - U.x0) < (unsigned)U.w);
    const size_t cell = map_base + (size_t)(r * a.mw + x);
    float* po = a.out + cell;
    uint8_t* pm = a.mask + cell;
    po = skip ? sink : po;
    pm = skip ? reinterpret_cast<uint8_t*>(sink + 8) : pm;
    *reinterpret_cast<float4*>(po) = make_float4(a.fill, a.fill, a.fill, a.fill);
    *reinterpret_cast<uint32_t*>(pm) = 0u;
  };
  auto publish_geometry = [&]() {
    if (publisher && threadIdx.x < strip::kMaxStrips)
      a.g_wins[(size_t)b * strip::kMaxStrips + threadIdx.x] = geom->win[threadIdx.x];
    if (publisher && threadIdx.x == 0) a.g_unions[b] = narrow16(U);
  };

  const bool glob = a.to_global != 0;
  const float p4 = fr[4], p5 = fr[5], p7 = fr[7], p8 = fr[8];
  const float y0 = glob ? fr[10] : 1.0f, y2r = glob ? fr[12] : 0.0f;
  const float y6 = glob ? fr[16] : 0.0f, y8 = glob ? fr[18] : 1.0f;
  const float cam_h = fr[9], tx = glob ? fr[19] : 0.0f, tz = glob ? fr[20] : 0.0f;
  const float wo = fr[21], ho = fr[22];
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
            bool ok = ux < (unsigned)w.w && uz < (unsigned)w.h && zz >= a.dmin && zz <= a.dmax;
            if (!LEAN) ok = ok && !__builtin_isunordered(xf, zf) && h1 <= a.hmax;
            const float sval = HAS_VALUE ? sv[u][k] : h1;
            if (HAS_VALUE) ok = ok && (sval == sval);            // NaN never replaces a number
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
  while (fs < fill_steps) fill_step();
  lds_barrier();
  if (area > 0) {
    // Flush: a group of the window inside this strip's cover goes straight to the map when no
    // other strip's cover holds it, else to the slab (k_strip_merge combines those).
    const int pid = (b * a.oc + chl) * nparts + part;
    float* slab = a.slabs + (size_t)pid * a.slab_stride;
    const int wg4 = w.w >> 2;
    const int groups = wg4 * w.h;
    const float wg4_inv = 1.0f / (float)wg4;
    for (int i = threadIdx.x; i < groups; i += kScatterThreads) {
      int row = (int)((float)i * wg4_inv);
      row -= (row * wg4 > i);
      row += ((row + 1) * wg4 <= i);
      const int x = w.x0 + ((i - row * wg4) << 2), z = w.z0 + row;
      const uint32_t* cv = covers + (z - U.z0) * nparts;
      if (!strip::in_cover(cv[part], x)) continue;
      bool shared = false;
      for (int p = 0; p < nparts; ++p) shared = shared || (p != part && strip::in_cover(cv[p], x));
      const float4 v = *reinterpret_cast<const float4*>(lds + i * 4);
      if (shared || a.out == nullptr) {
        *reinterpret_cast<float4*>(slab + i * 4) = v;
      } else {
        const size_t cell = map_base + (size_t)z * a.mw + x;
        *reinterpret_cast<float4*>(a.out + cell) = v;
        const uint32_t mk = (uint32_t)mask_of(v.x, a.fill) | ((uint32_t)mask_of(v.y, a.fill) << 8) |
                            ((uint32_t)mask_of(v.z, a.fill) << 16) |
                            ((uint32_t)mask_of(v.w, a.fill) << 24);
        *reinterpret_cast<uint32_t*>(a.mask + cell) = mk;
      }
    }
  }
  publish_geometry();
}

struct StripMergeArgs {
  int b0, oc, ch0, oc_total, mh, mw;
  int P, slab_stride, max_rows;
  float fill;
  const Win16* g_wins;
  const Win16* g_unions;
  const uint32_t* g_covers;
  const float* slabs;
  float* out;
  uint8_t* mask;
};

// One thread per float4 group of a frame's union window: groups in exactly one cover were
// written by k_strip_scatter; groups in none get the fill value; the others the max / min of
// the slabs of the strips covering them.
template <int RED>
__global__ void __launch_bounds__(kMergeThreads)
k_strip_merge(StripMergeArgs a) {
  const int fcl = blockIdx.y;                  // (frame of the launch) * oc + channel of the group
  const int bl = fcl / a.oc, b = a.b0 + bl;
  const int chl = fcl - bl * a.oc;
  const Window U = widen(a.g_unions[b]);
  const int ug4 = U.w >> 2;
  const int total = ug4 * U.h;
  const int i = blockIdx.x * kMergeThreads + threadIdx.x;
  if (i >= total) return;
  const int row = i / ug4;
  const int zb = U.z0 + row, x = U.x0 + ((i - row * ug4) << 2);
  const uint32_t* cv = a.g_covers + ((size_t)b * a.max_rows + row) * a.P;
  int count = 0;
  for (int p = 0; p < a.P; ++p) count += strip::in_cover(cv[p], x) ? 1 : 0;
  if (count == 1) return;
  float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
  if (count >= 2) {
    const float* slabs = a.slabs + ((size_t)(b * a.oc + chl) * a.P) * a.slab_stride;
    for (int p = 0; p < a.P; ++p) {
      if (!strip::in_cover(cv[p], x)) continue;
      const Window w = widen(a.g_wins[(size_t)b * strip::kMaxStrips + p]);
      const float4 s = *reinterpret_cast<const float4*>(
          slabs + (size_t)p * a.slab_stride + (size_t)(zb - w.z0) * w.w + (x - w.x0));
      acc.x = combine<RED>(acc.x, s.x);
      acc.y = combine<RED>(acc.y, s.y);
      acc.z = combine<RED>(acc.z, s.z);
      acc.w = combine<RED>(acc.w, s.w);
    }
  }
  const size_t fo = (size_t)b * a.oc_total + a.ch0 + chl;
  const size_t cell = fo * (size_t)a.mh * a.mw + (size_t)zb * a.mw + x;
  *reinterpret_cast<float4*>(a.out + cell) = acc;
  const uint32_t mk = (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
                      ((uint32_t)mask_of(acc.z, a.fill) << 16) |
                      ((uint32_t)mask_of(acc.w, a.fill) << 24);
  *reinterpret_cast<uint32_t*>(a.mask + cell) = mk;
}

// Test hook kernel: the geometry of every frame exactly as k_strip_scatter derives it.
__global__ void __launch_bounds__(64)
k_strip_geometry_dump(strip::Cfg cfg, const float* frames, strip::FrameGeom* out) {
  __shared__ strip::FrameGeom g;
  float fr[23];
  for (int i = 0; i < 23; ++i) fr[i] = frames[(size_t)blockIdx.x * 32 + i];
  strip_geometry_wave(cfg, fr, (int)threadIdx.x, &g);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = g;
}

}  // namespace
}  // namespace dm
