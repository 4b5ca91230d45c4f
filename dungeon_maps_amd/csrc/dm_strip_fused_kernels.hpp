// k_strip_fused: dm_orth_project_fused_f32 (a batch projected straight into ONE map: MapBuilder's
// running world map for maps that share a frame, maps.py:2471-2508 -> 2181-2287; BASELINE
// configs[3]) on column strips.
//
// The per-frame maps are never written, so nothing has to be owned, filled or combined per frame:
// a workgroup takes ONE column strip of a GROUP of up to eight consecutive frames and reduces all
// their pixels into one LDS window -- the bounding box of the frames' windows for that strip,
// which for the frames of a trajectory is hardly larger than one frame's -- and flushes it as a
// slab; k_fuse_windows (dm_window_kernels.hpp) then reduces the slabs covering each cell of the
// map.  Against one workgroup per (frame, strip) that is a fraction of the slabs to write, and
// of the windows the fuse kernel has to visit per cell.  Frames with unrelated poses simply get
// groups of one.
//
// Same pixel arithmetic as k_strip_scatter (dm_strip_kernels.hpp; the contract of dm_pixel.hpp).
#pragma once

#include "dm_strip_kernels.hpp"

namespace dm {
namespace {

constexpr int kFusedSlots = 5;          // image rows of a thread in flight per pipeline stage
constexpr int kFusedMaxGroup = 8;       // frames per workgroup: one wave derives their windows (8 frames x 8 corners)

struct FusedArgs {
  int W, H;
  int clip, flip_h;
  float cx, cy, fx, fy, res;
  float fx_inv, fy_inv, res_inv;
  float dmin, dmax, hmax;
  float Hm1, mhm1;
  float p4, p5, p7, p8;       // the batch's pitch rotation
  int wp, P;                  // strip width (pixels, multiple of 4), strips
  int F;                      // frames per group
  int nb;                     // frames of this launch (poses[0 .. nb))
  int group0;                 // index of the launch's first group among the call's
  int dc, valid_c;
  int slab_stride;            // cells of the LDS window region = of a slab
  int max_rows;               // rows the span table holds (>= the group window's height)
  int mh, mw;
  float fill;
  float cam_h;                // DEFER: the camera height all frames share (added at the flush)
  float inv, reach, g0, g1;   // cells per metre; strip::Cfg::reach; z1 = g * depth at the extreme live rows
  int cone_ok;
  const float* depth;         // the launch's first frame
  const uint8_t* valid;
  float* slabs;               // ((group * dc + ch) * P + strip) * slab_stride
  Win16* wins;                // (groups, P)
  uint32_t* spans;            // (groups, P, max_rows)   lo | hi << 16 of every window row: the cells the slab holds
  int* status;
#ifdef DM_STAMPS
  long long* stamps;
#endif
  StripPose poses[kPoseFrames];
};

// what strip::pose_of / window_of read of a configuration
struct FusedRig { int mw, mh, flip_h, cone_ok; float inv, reach; };

// what wave 0 leaves of the group's frames for the span pass: each frame's own window and cone edges for this strip
struct FusedFrames {
  Win16 win[kFusedMaxGroup];
  strip::Line L[kFusedMaxGroup], R[kFusedMaxGroup];
};

// LDS, in floats: [window region: slab_stride | 64 scratch cells | ray slopes of the image rows: H | the group's
// window | the frames' windows and edges | the rows' spans: max_rows]
__host__ __device__ inline size_t fused_lds_bytes(int slab_cells, int H, int max_rows) {
  return ((size_t)slab_cells + 64 + (size_t)((H + 3) & ~3) + 16 + (size_t)max_rows) * 4 + sizeof(FusedFrames);
}

// RED: kMin / kMax.  HAS_VALID / LEAN as in k_strip_scatter.  DEFER (LEAN only: no height
// truncation to test y1 against): every frame has the same camera height, which is then added
// when the window is flushed instead of per pixel (x -> RN(x + c) is monotone).
template <int RED, bool HAS_VALID, bool LEAN, bool DEFER>
__global__ void __launch_bounds__(kScatterThreads)
k_strip_fused(FusedArgs a) {
  constexpr int VEC = 4, S = kFusedSlots;
  static_assert(LEAN || !DEFER, "the height truncation needs the camera height per pixel");
  extern __shared__ float lds[];
  const int part = blockIdx.x, ch = blockIdx.y, gl = blockIdx.z;
  const int f0 = gl * a.F;                                   // the group's first frame, within the launch
  const int nf = min(a.F, a.nb - f0);
  const int q0 = part * a.wp;
  int q1 = q0 + a.wp; if (q1 > a.W) q1 = a.W;
  const int nx = (q1 - q0 + VEC - 1) / VEC;
  const int ntx = nx < kScatterThreads ? nx : kScatterThreads;
  const int rows_per_iter = kScatterThreads / ntx;
  const int gx = threadIdx.x % ntx, gy = threadIdx.x / ntx;
  const int nslots = (a.H + rows_per_iter - 1) / rows_per_iter;        // rows of the image a thread visits
  const int cpf = (nslots + S - 1) / S;                                // pipeline stages per frame
  const size_t N = (size_t)a.H * a.W;
  const int fstride = a.dc * a.H;                            // image rows from one frame to the next
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  // the group's frames of this channel as ONE raw buffer (frames are dc * N elements apart)
  const __amdgpu_buffer_rsrc_t rs_depth = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.depth) + ((size_t)f0 * a.dc + ch) * N, 0,
      (unsigned)(((size_t)(nf - 1) * a.dc * N + N) * 4u), 0x00020000);
  const int vstride = a.valid_c * a.H;
  const __amdgpu_buffer_rsrc_t rs_valid = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(HAS_VALID ? a.valid + ((size_t)f0 * a.valid_c + (a.valid_c == 1 ? 0 : ch)) * N : a.valid), 0,
      HAS_VALID ? (unsigned)((size_t)(nf - 1) * a.valid_c * N + N) : 0u, 0x00020000);
  const float qnan = __builtin_nanf("");
  const bool col_live = gx < nx;
  const int q = q0 + gx * VEC;
  float za[S][VEC], zb[S][VEC], aya[S], ayb[S];
  // rows of stage (f, c): slots c * S + u, u < S, of frame f.  No branch: rows past the image (inside
  // a slot, or whole slots of a frame's last stage) repeat its last row and a stage past the group
  // repeats the last frame -- max / min are idempotent, and straight-line code lets the S rows'
  // instruction streams interleave.
  auto load_stage = [&](float (&z)[S][VEC], int f, int c) {
    // (the pipeline requests one stage more than the group has -- never projected: its loads get an offset
    // past the end of the buffer and move no bytes.  Until round 5 they repeated the last frame's rows: a
    // fifth stage of loads for the four of a one-frame group, 100 MB fetched for 79 MB of depth maps at cfg4)
    const bool past = f >= nf;
    f = past ? nf - 1 : f;
#pragma unroll
    for (int u = 0; u < S; ++u) {
      int rr = (c * S + u) * rows_per_iter + gy;
      rr = rr < a.H ? rr : a.H - 1;
      const int at = __mul24(f * fstride + rr, a.W) + q;
      const f32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_depth, past ? 0x7ffffff0 : at << 2, 0, 0);
      z[u][0] = t.x; z[u][1] = t.y; z[u][2] = t.z; z[u][3] = t.w;
      if (HAS_VALID) {
        const unsigned ok4 = __builtin_amdgcn_raw_buffer_load_b32(rs_valid, past ? 0x7ffffff0 : __mul24(f * vstride + rr, a.W) + q, 0, 0);
#pragma unroll
        for (int k = 0; k < VEC; ++k) z[u][k] = ((ok4 >> (8 * k)) & 0xffu) ? z[u][k] : qnan;
      }
    }
  };
  if (col_live) load_stage(za, 0, 0);            // the first rows: kernel arguments only
#ifdef DM_STAMPS
  long long stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DM_STAMP(0);

  float* aytab = lds + a.slab_stride + 64;
  int* gwin = reinterpret_cast<int*>(aytab + ((a.H + 3) & ~3));     // {x0 | z0 << 16, w | h << 16, inside, ok}
  uint32_t* rowspan = reinterpret_cast<uint32_t*>(gwin + 16);       // [max_rows]
  FusedFrames* frames_lds = reinterpret_cast<FusedFrames*>(rowspan + a.max_rows);
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  typedef const __attribute__((address_space(4))) float cfloat;
  cfloat* const poses = (cfloat*)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                  offsetof(FusedArgs, poses)) + f0 * kPoseFloats;
  if (wave == 0) {
    // The group's window: lane = frame x corner derives the frames' windows for this strip (the
    // calls of strip::frame_geometry), their bounding box is the workgroup's window.
    using namespace strip;
    const int fr = min(lane >> 3, nf - 1), k = lane & 7;
    cfloat* const pf = poses + fr * kPoseFloats;        // (per-lane loads from the kernel arguments)
    const float y0 = pf[0], y2 = pf[1], y6 = pf[2], y8 = pf[3], tx = pf[4], tz = pf[5], wo = pf[6], ho = pf[7];
    FusedRig c;
    c.mw = a.mw; c.mh = a.mh; c.flip_h = a.flip_h; c.cone_ok = a.cone_ok; c.inv = a.inv; c.reach = a.reach;
    // the strip's live columns and their ray slopes (dm_strip.hip make_plan)
    int lq0 = q0 < a.clip ? a.clip : q0, lq1 = q1 > a.W - a.clip ? a.W - a.clip : q1;
    const bool live = lq0 < lq1;
    const float ax_lo = (float)(((double)lq0 - (double)a.cx) / (double)a.fx);
    const float ax_hi = (float)(((double)(lq1 - 1) - (double)a.cx) / (double)a.fx);
    const float d = (k & 1) ? a.dmax : a.dmin;
    const float ax = (k & 2) ? ax_hi : ax_lo;
    const float gg = (k & 4) ? a.g1 : a.g0;
    float ccx = ax * d * c.inv, ccz = gg * d * c.inv;               // strip_corners
    ccx = live ? ccx : 0.0f; ccz = live ? ccz : 0.0f;
    const Pose p = pose_of(c, y0, y2, y6, y8, tx, tz, wo, ho);
    const float xf = p.y0 * ccx + p.y6 * ccz + p.xd;                // strip_geometry
    const float zf = p.y2 * ccx + p.y8 * ccz + p.zd;
    const float lx = min8(xf), hx = max8(xf), lz = min8(zf), hz = max8(zf);
    bool in = false;
    const Win16 ww = window_of(c, lx, hx, lz, hz, p.slack, in);
    const bool on = live & (p.ok != 0);
    const bool some = on & (ww.w > 0);
    {   // the frame's cone edges for this strip (strip::cfg_rig's tmin / tmax, strip::strip_geometry's edges)
      const float t0 = ax_lo / a.g0, t1 = ax_lo / a.g1, t2 = ax_hi / a.g0, t3 = ax_hi / a.g1;
      const float tmin = fmin2(fmin2(t0, t1), fmin2(t2, t3)), tmax = fmax2(fmax2(t0, t1), fmax2(t2, t3));
      const Line l = cone_edge(c, p, tmin, true), r = cone_edge(c, p, tmax, false);
      if (k == 0 && (lane >> 3) < nf) {
        frames_lds->win[lane >> 3] = some ? ww : Win16{0, 0, 0, 0};
        frames_lds->L[lane >> 3] = some ? l : Line{0.0f, 0.0f, 0.0f, 0.0f};
        frames_lds->R[lane >> 3] = some ? r : Line{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
    int x0 = 32767, x1 = 0, z0 = 32767, z1 = 0, all_in = 1, all_ok = 1;
    for (int f = 0; f < nf; ++f) {           // (wave-uniform: the frames' results out of lanes 8 f)
      const int s_ = __builtin_amdgcn_readlane((int)some, 8 * f);
      const int wx0 = __builtin_amdgcn_readlane((int)ww.x0, 8 * f), wz0 = __builtin_amdgcn_readlane((int)ww.z0, 8 * f);
      const int wxw = __builtin_amdgcn_readlane((int)ww.w, 8 * f), wzh = __builtin_amdgcn_readlane((int)ww.h, 8 * f);
      all_in &= __builtin_amdgcn_readlane((int)(on & in & (ww.w > 0)), 8 * f);
      all_ok &= __builtin_amdgcn_readlane(live ? p.ok : 1, 8 * f);
      if (s_) { x0 = min(x0, wx0); x1 = max(x1, wx0 + wxw); z0 = min(z0, wz0); z1 = max(z1, wz0 + wzh); }
    }
    if (lane == 0) {
      const bool any = x1 > x0;
      gwin[0] = any ? (x0 & 0xffff) | (z0 << 16) : 0;
      gwin[1] = any ? ((x1 - x0) & 0xffff) | ((z1 - z0) << 16) : 0;
      gwin[2] = all_in; gwin[3] = all_ok;
    }
  } else {
    // the window region of LDS and the ray-slope table (maps.py:670-678; border rows poisoned)
    constexpr int kInitThreads = kScatterThreads - 64;
    const int t = (int)threadIdx.x - 64;
    const float lds_init = DEFER ? (RED == kMax ? -INFINITY : INFINITY) : a.fill;
    for (int i = t * 4; i < a.slab_stride + 64; i += kInitThreads * 4)
      *reinterpret_cast<float4*>(lds + i) = make_float4(lds_init, lds_init, lds_init, lds_init);
    for (int r = t; r < a.H; r += kInitThreads) {
      float yr = (float)r;
      yr = a.flip_h ? a.Hm1 - yr : yr;
      float ay = div_markstein(yr - a.cy, a.fy, a.fy_inv);
      if (!LEAN) ay = (r < a.clip || r >= a.H - a.clip) ? qnan : ay;
      aytab[r] = ay;
    }
  }
  DM_STAMP(1);
  lds_barrier();
  DM_STAMP(2);
  Window w;
  {
    const int g0_ = __builtin_amdgcn_readfirstlane(gwin[0]), g1_ = __builtin_amdgcn_readfirstlane(gwin[1]);
    w.x0 = (short)(g0_ & 0xffff); w.z0 = (short)(g0_ >> 16); w.w = (short)(g1_ & 0xffff); w.h = (short)(g1_ >> 16);
  }
  const int inside = __builtin_amdgcn_readfirstlane(gwin[2]);
  // a group whose window does not fit what the host sized the launch for (cannot happen: the host
  // bounds the group's window from these very poses), or a frame the cone model does not hold
  // for: flagged, nothing projected
  const bool fits = __builtin_amdgcn_readfirstlane(gwin[3]) != 0 && w.w * w.h <= a.slab_stride && w.h <= a.max_rows;
  if (!fits) {
    if (threadIdx.x == 0 && a.status && (w.w > 0 || !__builtin_amdgcn_readfirstlane(gwin[3])))
      raise_status(a.status, kStatusFrameDidNotFit);
    w.w = 0; w.h = 0;
  }
  const int area = w.w * w.h;
  const int group = a.group0 + gl;
  if (threadIdx.x == 0) a.wins[(size_t)group * a.P + part] = area > 0 ? narrow16(w) : Win16{0, 0, 0, 0};
  if (area == 0) return;                  // (workgroup-uniform)

  const float flip_s = a.flip_h ? -1.0f : 1.0f, flip_c = a.flip_h ? a.mhm1 : 0.0f;
  typedef __attribute__((address_space(3))) float lds_float;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_float*)lds;
  const unsigned dummy = lds_base + (((unsigned)a.slab_stride + (unsigned)lane) << 2);
  const int origin = (int)lds_base - 4 * (w.z0 * w.w + w.x0);
  auto lds_at = [](unsigned addr) { return (lds_float*)(uintptr_t)addr; };
  float ax[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    const float d = (float)(q + k) - a.cx;
    ax[k] = div_markstein(d, a.fx, a.fx_inv);
    if (!LEAN) ax[k] = (q + k < a.clip || q + k >= a.W - a.clip) ? qnan : ax[k];
  }
  auto load_ay = [&](float (&ay)[S], int c) {
#pragma unroll
    for (int u = 0; u < S; ++u) {
      int rr = (c * S + u) * rows_per_iter + gy;
      rr = rr < a.H ? rr : a.H - 1;
      ay[u] = aytab[rr];
    }
  };
  // one frame's pose as wave-uniform values
  struct PoseS { float y0, y2, y6, y8, tx, tz, wo, ho, cam_h; };
  auto load_pose = [&](int f) {
    cfloat* pr = poses + (f < nf ? f : nf - 1) * kPoseFloats;
    PoseS o;
    o.y0 = pr[0]; o.y2 = pr[1]; o.y6 = pr[2]; o.y8 = pr[3]; o.tx = pr[4]; o.tz = pr[5]; o.wo = pr[6]; o.ho = pr[7];
    o.cam_h = pr[8];
    return o;
  };
  auto project_stage = [&](auto tested, const float (&z)[S][VEC], const float (&ayr)[S], const PoseS& ps, int c) {
    constexpr bool kTest = decltype(tested)::value;
    constexpr bool kExecMask = LEAN && DEFER;
    unsigned long long exec_all = 0;
    if (kExecMask) asm volatile("s_mov_b64 %0, exec" : "=s"(exec_all));
    const float p4 = a.p4, p5 = a.p5, p7 = a.p7, p8 = a.p8;
#pragma unroll
    for (int u = 0; u < S; ++u) {
      const float ay = ayr[u];
      unsigned li[VEC];
      float hv[VEC], xfv[VEC], zfv[VEC];
      typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int k = 0; k < VEC; k += 2) {
        const f2 zz = {z[u][k], z[u][k + 1]};
        const f2 axp = {ax[k], ax[k + 1]};
        const f2 X = axp * zz;
        const f2 Y = zz * ay;                                            // maps.py:677-678
        f2 h1 = __builtin_elementwise_fma(zz, (f2){p7, p7}, Y * p4);                  // maps.py:790-797
        if (!DEFER) h1 = h1 + ps.cam_h;
        const f2 z1 = __builtin_elementwise_fma(zz, (f2){p8, p8}, Y * p5);
        const f2 x2 = __builtin_elementwise_fma(z1, (f2){ps.y6, ps.y6}, X * ps.y0) + ps.tx;      // maps.py:884-892
        const f2 z2 = __builtin_elementwise_fma(z1, (f2){ps.y8, ps.y8}, X * ps.y2) + ps.tz;
        const f2 ri = {a.res_inv, a.res_inv}, nres = {-a.res, -a.res};
        const f2 qx = x2 * ri, qz = z2 * ri;                             // exact division
        f2 xf2 = __builtin_elementwise_fma(__builtin_elementwise_fma(nres, qx, x2), ri, qx) + ps.wo;
        f2 zf2 = __builtin_elementwise_fma(__builtin_elementwise_fma(nres, qz, z2), ri, qz) + ps.ho;
        zf2 = __builtin_elementwise_fma(zf2, (f2){flip_s, flip_s}, (f2){flip_c, flip_c});
        xf2 = xf2 + 0.5f;
        zf2 = zf2 + 0.5f;
        xfv[k] = xf2.x; xfv[k + 1] = xf2.y;
        zfv[k] = zf2.x; zfv[k + 1] = zf2.y;
        hv[k] = h1.x; hv[k + 1] = h1.y;
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float zz = z[u][k];
        const int ix = floor_to_int(xfv[k]), iz = floor_to_int(zfv[k]);
        bool ok = (zz >= a.dmin) & (zz <= a.dmax);            // maps.py:537-544
        if (kTest) ok = ok & ((unsigned)(ix - w.x0) < (unsigned)w.w) & ((unsigned)(iz - w.z0) < (unsigned)w.h);
        if (!LEAN) ok = ok & !__builtin_isunordered(xfv[k], zfv[k]) & (hv[k] <= a.hmax);     // maps.py:286-288
        unsigned addr = (unsigned)(((__mul24(iz, w.w) + ix) << 2) + origin);
        asm("" : "+v"(addr));
        li[k] = ok ? addr : dummy;
        if (kExecMask && !kTest) li[k] = addr;
      }
      if (kExecMask && !kTest) {
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(li[0] == li[VEC - 1]) != 0, 0)) {
          // rare: some thread's four pixels may share a cell -- the plain way, validity as a value
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const bool ok = (z[u][k] >= a.dmin) & (z[u][k] <= a.dmax);
            li[k] = ok ? li[k] : dummy;
          }
#pragma unroll
          for (int k = 0; k + 1 < VEC; ++k) {
            const bool same = li[k] == li[k + 1];
            const float m = combine<RED>(hv[k], hv[k + 1]);
            hv[k + 1] = same ? m : hv[k + 1];
            li[k] = same ? dummy : li[k];
          }
#pragma unroll
          for (int k = 0; k < VEC; ++k) lds_reduce<RED>(lds_at(li[k]), hv[k]);
        } else {
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            if (RED == kMax)
              asm volatile("v_cmpx_le_f32_e32 vcc, %0, %2\n\tv_cmpx_ge_f32_e32 vcc, %1, %2\n\t"
                           "ds_max_f32 %3, %4\n\ts_mov_b64 exec, %5"
                           :: "s"(a.dmin), "s"(a.dmax), "v"(z[u][k]), "v"(li[k]), "v"(hv[k]), "s"(exec_all)
                           : "vcc", "memory");
            else
              asm volatile("v_cmpx_le_f32_e32 vcc, %0, %2\n\tv_cmpx_ge_f32_e32 vcc, %1, %2\n\t"
                           "ds_min_f32 %3, %4\n\ts_mov_b64 exec, %5"
                           :: "s"(a.dmin), "s"(a.dmax), "v"(z[u][k]), "v"(li[k]), "v"(hv[k]), "s"(exec_all)
                           : "vcc", "memory");
          }
        }
        continue;
      }
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(li[0] == li[VEC - 1] && li[0] != dummy) != 0, 0)) {
#pragma unroll
        for (int k = 0; k + 1 < VEC; ++k) {
          const bool same = li[k] == li[k + 1];
          const float m = combine<RED>(hv[k], hv[k + 1]);
          hv[k + 1] = same ? m : hv[k + 1];
          li[k] = same ? dummy : li[k];
        }
      }
#pragma unroll
      for (int k = 0; k < VEC; ++k) lds_reduce<RED>(lds_at(li[k]), hv[k]);
    }
  };
  // The pipeline over the stages of all frames of the group: while stage t is projected, the
  // rows of stage t + 1 (the same frame's next slots, or the next frame's first) are in flight.
  auto pipeline = [&](auto tested) {
    if (!col_live) return;
    const int T = nf * cpf;
    int lf = 0, lc = 0;                    // the stage whose rows are requested next
    auto next = [&](int& f, int& c) { if (++c == cpf) { c = 0; ++f; } };
    next(lf, lc);                          // (stage 0 is in flight since the kernel's start)
    load_ay(aya, 0);
    PoseS cur = load_pose(0), nxt = load_pose(1);
    int pf = 0, pc = 0;                    // the stage projected next
    if (wave >= 12) __builtin_amdgcn_s_setprio(3);
    else if (wave >= 8) __builtin_amdgcn_s_setprio(2);
    else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
    for (int t = 0; t < T; t += 2) {
      load_stage(zb, lf, lc); load_ay(ayb, lc); next(lf, lc);
      project_stage(tested, za, aya, cur, pc);
      next(pf, pc);
      if (pc == 0) { cur = nxt; nxt = load_pose(pf + 1); }
      if (t + 1 < T) {
        load_stage(za, lf, lc); load_ay(aya, lc); next(lf, lc);
        project_stage(tested, zb, ayb, cur, pc);
        next(pf, pc);
        if (pc == 0) { cur = nxt; nxt = load_pose(pf + 1); }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  DM_STAMP(3);
  if (inside) pipeline(std::false_type{}); else pipeline(std::true_type{});
  DM_STAMP(4);
  // The rows' spans: on map row z the group's pixels can only land between its frames' cone edges
  // (strip::row_cover, the cover of k_strip_scatter's strips) -- a fraction of the window's
  // bounding box, which is all the slab has to hold and all k_fuse_windows has to visit.  Eight
  // lanes per row, lane = frame (a group's last frame repeated), the hull over the frames by DPP
  // moves; a wave does this as soon as it leaves the pixel loop, under the slower waves' last
  // rows.  (Inside its frame's window by construction: the window's x range is a multiple of 4.)
  {
    const int fq = min(lane & 7, nf - 1);
    const Win16 fw = frames_lds->win[fq];
    const strip::Line fl = frames_lds->L[fq], fr = frames_lds->R[fq];
    uint32_t* const gspans = a.spans + ((size_t)group * a.P + part) * a.max_rows;
    for (int r0 = 0; r0 < w.h; r0 += kScatterThreads / 8) {       // (uniform trips: the DPP moves)
      const int r = r0 + ((int)threadIdx.x >> 3);
      const uint32_t cov = r < w.h ? strip::row_cover(fw, fl, fr, w.z0 + r, a.mw) : 0u;
      int lo = cov ? (int)(cov & 0xffffu) : 32767, hi = (int)(cov >> 16);
      lo = min(lo, dpp_i<0xB1>(lo)); lo = min(lo, dpp_i<0x4E>(lo)); lo = min(lo, dpp_i<0x141>(lo));
      hi = max(hi, dpp_i<0xB1>(hi)); hi = max(hi, dpp_i<0x4E>(hi)); hi = max(hi, dpp_i<0x141>(hi));
      if ((lane & 7) == 0 && r < w.h) {
        const uint32_t span = hi > lo ? (uint32_t)lo | ((uint32_t)hi << 16) : 0u;
        rowspan[r] = span;
        if (ch == 0) gspans[r] = span;
      }
    }
  }
  DM_STAMP(5);
  lds_barrier();
  DM_STAMP(6);
  // flush: the spans of the window's rows -> the workgroup's slab, 16 lanes per row
  float* slab = a.slabs + (((size_t)group * a.dc + ch) * a.P + part) * a.slab_stride;
  const int l16 = (int)threadIdx.x & 15;
  for (int row = (int)threadIdx.x >> 4; row < w.h; row += kScatterThreads / 16) {
    const uint32_t span = rowspan[row];
    const int lo = (int)(span & 0xffffu), hi = (int)(span >> 16);
    const int cell0 = row * w.w - w.x0;
    for (int x = lo + (l16 << 2); x < hi; x += 64) {
      float4 v = *reinterpret_cast<const float4*>(lds + cell0 + x);
      if (DEFER) { v.x += a.cam_h; v.y += a.cam_h; v.z += a.cam_h; v.w += a.cam_h; }
      *reinterpret_cast<float4*>(slab + cell0 + x) = v;
    }
  }
  DM_STAMP(7);
  DM_STAMPS_OUT();
}

}  // namespace
}  // namespace dm
