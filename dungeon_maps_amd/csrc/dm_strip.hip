// Strip path of orth_project (max / min): host side.  See dm_strip_geometry.hpp for the
// geometry and dm_strip_kernels.hpp for the kernels.
//
// What the host does per call is pose independent: it validates the frame records (the
// rotations have the axis-aligned pattern the fast arithmetic needs, one camera pitch for the
// whole batch, magnitudes within the float32 slack), copies them to the device (or takes
// them from the device as they are: dm_frames_prepare_f32 / dm_orth_project_prepared_f32)
// and launches k_strip_scatter + k_strip_merge with sizes taken from a bound that holds for
// EVERY yaw and position of the camera (cached per camera rig).  The windows themselves are
// derived on the device.
#include <math.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "dm_kernels.hpp"
#include "dm_strip_kernels.hpp"

namespace dm {

void note_split(int pc, int pr, int pd, int path);     // dm_window.hip (dm_debug_last_split)

namespace {

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// Pose-independent plan of a call.
struct Plan {
  strip::Cfg cfg;
  int P, wp;
  float res_inv, fx_inv, fy_inv;
  bool lean;
};

// Sizes that hold for every camera yaw / position (same pitch): what the launch is sized with.
struct RigBound {
  dm_params key;
  float pitch[4];               // Rp[4], Rp[5], Rp[7], Rp[8]
  int slack;                    // cells of slack the bound was computed for
  int P;
  bool valid, fits;
  int slab_stride;              // cells of the largest window any strip can have
  int max_rows;                 // rows of the largest union window
  int max_union;                // cells of the largest union window
};

thread_local int g_force_strips = 0;       // dm_debug_force_strips

bool make_plan(const dm_params& p, Plan& plan) {
  if (p.reduction != DM_REDUCE_MAX && p.reduction != DM_REDUCE_MIN) return false;
  if (p.mw % 4 != 0 || p.W % 4 != 0) return false;
  if (p.mw > 32767 || p.mh > 32767) return false;
  if (!(p.fill == p.fill)) return false;
  if (!p.has_dmin || !p.has_dmax || !(p.dmin >= 0.0f) || !(p.dmax >= p.dmin) || !isfinite(p.dmax))
    return false;
  Parts parts = choose_parts(p, 1, 1);
  if (g_force_strips > 0) {      // tests: this many column strips whatever the cost model says
    parts.pc = g_force_strips; parts.pr = 1; parts.pd = 1;
    parts.wp = ((p.W + parts.pc - 1) / parts.pc + 3) & ~3;
    parts.hp = p.H;
  }
  if (parts.pr != 1 || parts.pd != 1 || parts.pc > strip::kMaxStrips || parts.wp % 4 != 0) return false;
  if (!exact_reciprocal(p.res, &plan.res_inv) || !exact_reciprocal(p.fx, &plan.fx_inv) ||
      !exact_reciprocal(p.fy, &plan.fy_inv) || p.res < 1e-6f || p.res > 1e6f || p.fx < 1e-6f ||
      p.fx > 1e6f || p.fy < 1e-6f || p.fy > 1e6f)
    return false;
  plan.P = parts.pc;
  plan.wp = parts.wp;
  strip::Cfg& c = plan.cfg;
  memset(&c, 0, sizeof(c));
  c.P = plan.P; c.mw = p.mw; c.mh = p.mh; c.flip_h = p.flip_h != 0; c.to_global = p.to_global != 0;
  c.dmin = p.dmin; c.dmax = p.dmax;
  c.res_inv = 1.0 / (double)p.res;
  const int clip = p.clip_border > 0 ? p.clip_border : 0;
  int r0 = clip, r1 = p.H - clip;
  if (r0 >= r1) return false;
  double ay[2];
  const int rs[2] = {r0, r1 - 1};
  for (int i = 0; i < 2; ++i) {
    double yr = rs[i];
    if (p.flip_h) yr = (double)(p.H - 1) - yr;
    ay[i] = (yr - (double)p.cy) / (double)p.fy;
  }
  c.ay_lo = ay[0] < ay[1] ? ay[0] : ay[1];
  c.ay_hi = ay[0] < ay[1] ? ay[1] : ay[0];
  bool any = false;
  for (int s = 0; s < plan.P; ++s) {
    int q0 = s * plan.wp, q1 = q0 + plan.wp < p.W ? q0 + plan.wp : p.W;
    if (q0 < clip) q0 = clip;
    if (q1 > p.W - clip) q1 = p.W - clip;
    c.live[s] = q0 < q1;
    if (!c.live[s]) continue;
    any = true;
    c.ax_lo[s] = ((double)q0 - (double)p.cx) / (double)p.fx;
    c.ax_hi[s] = ((double)(q1 - 1) - (double)p.cx) / (double)p.fx;
  }
  if (!any) return false;
  plan.lean = !p.valid_c && isfinite(p.dmin) && !p.has_hmax && p.clip_border <= 0;
  return true;
}

// The frame records the strip path can take: axis-aligned rotations (the fast arithmetic),
// one pitch for the batch, finite values; returns the cells of slack that cover every frame
// (>= what the device computes for it) or -1.
int validate_frames(const dm_params& p, const strip::Cfg& c, const dm_frame* f, int B) {
  double worst = 0.0;
  const double inv = c.res_inv;
  for (int b = 0; b < B; ++b) {
    const float* rp = f[b].Rp;
    const float* ry = f[b].Ry;
    if (!(rp[0] == 1.0f && rp[1] == 0.0f && rp[2] == 0.0f && rp[3] == 0.0f && rp[6] == 0.0f)) return -1;
    if (p.to_global &&
        !(ry[1] == 0.0f && ry[3] == 0.0f && ry[4] == 1.0f && ry[5] == 0.0f && ry[7] == 0.0f))
      return -1;
    if (rp[4] != f[0].Rp[4] || rp[5] != f[0].Rp[5] || rp[7] != f[0].Rp[7] || rp[8] != f[0].Rp[8]) return -1;
    double m = fabs((double)f[b].width_offset) + fabs((double)f[b].height_offset) + (double)p.mh;
    double rot = fabs((double)rp[4]) + fabs((double)rp[5]) + fabs((double)rp[7]) + fabs((double)rp[8]) + 2.0;
    if (p.to_global) {
      m += 2.0 * (fabs((double)f[b].tx) + fabs((double)f[b].tz)) * inv;
      rot *= fabs((double)ry[0]) + fabs((double)ry[2]) + fabs((double)ry[6]) + fabs((double)ry[8]) + 1.0;
    }
    m = 2.0 * m + (double)p.dmax * inv * rot * 2.0 + fabs((double)f[b].cam_height) * inv;
    if (!(m == m) || !isfinite(m)) return -1;
    if (m > worst) worst = m;
  }
  const double slack = 2.0 + 8.0 * worst * (1.0 / 8388608.0);
  if (slack > 16.0) return -1;
  return (int)ceil(slack);
}

// Bound of the window / union sizes over every yaw (and any position): the truncated cones are
// rotated in 1440 steps; between two steps an extent grows by at most Rmax * dtheta.
void compute_bound(const dm_params& p, const Plan& plan, const dm_frame& f0, int slack, RigBound& rb) {
  const strip::Cfg& c = plan.cfg;
  rb.key = p; rb.slack = slack; rb.P = plan.P; rb.valid = true; rb.fits = false;
  rb.pitch[0] = f0.Rp[4]; rb.pitch[1] = f0.Rp[5]; rb.pitch[2] = f0.Rp[7]; rb.pitch[3] = f0.Rp[8];
  // the cone model must hold (every row looks forward): checked with a neutral pose
  float rec[23] = {0};
  memcpy(rec, f0.Rp, 9 * sizeof(float));
  rec[9] = f0.cam_height;
  rec[10] = 1.0f; rec[14] = 1.0f; rec[18] = 1.0f;
  rec[21] = (float)(p.mw / 2); rec[22] = (float)(p.mh / 2);
  strip::Cfg c0 = c;
  c0.to_global = 0;
  const strip::Affine a = strip::frame_affine_f(c0, rec);
  if (!strip::cone_basis(c0, a).ok) return;
  // corner vectors in the camera's local frame, in cells (the yaw rotates them rigidly)
  std::vector<double> cx, cz;
  std::vector<int> owner;
  double rmax = 0.0;
  for (int s = 0; s < plan.P; ++s) {
    if (!c.live[s]) continue;
    for (int k = 0; k < 8; ++k) {
      double xf, zf;
      strip::cone_corner(c0, a, c.ax_lo[s], c.ax_hi[s], k, xf, zf);
      xf -= a.xd; zf -= a.zd;
      cx.push_back(xf); cz.push_back(zf); owner.push_back(s);
      const double r = sqrt(xf * xf + zf * zf);
      if (r > rmax) rmax = r;
    }
  }
  if (!isfinite(rmax)) return;
  const int steps = 1440;
  const double dtheta = 2.0 * M_PI / steps;
  const double lip = rmax * dtheta;
  const double pad_w = lip + 2.0 * slack + 9.0, pad_h = lip + 2.0 * slack + 3.0;
  double area = 0.0, uw = 0.0, uh = 0.0;
  for (int i = 0; i < steps; ++i) {
    const double cs = cos(i * dtheta), sn = sin(i * dtheta);
    double ulx = INFINITY, uhx = -INFINITY, ulz = INFINITY, uhz = -INFINITY;
    for (int s = 0; s < plan.P; ++s) {
      double lx = INFINITY, hx = -INFINITY, lz = INFINITY, hz = -INFINITY;
      for (size_t j = 0; j < cx.size(); ++j) {
        if (owner[j] != s) continue;
        const double x = cs * cx[j] + sn * cz[j], z = -sn * cx[j] + cs * cz[j];
        lx = x < lx ? x : lx; hx = x > hx ? x : hx; lz = z < lz ? z : lz; hz = z > hz ? z : hz;
      }
      if (!(hx >= lx)) continue;
      double w = hx - lx + pad_w, h = hz - lz + pad_h;
      if (w > p.mw) w = p.mw;
      if (h > p.mh) h = p.mh;
      if (w * h > area) area = w * h;
      ulx = lx < ulx ? lx : ulx; uhx = hx > uhx ? hx : uhx; ulz = lz < ulz ? lz : ulz; uhz = hz > uhz ? hz : uhz;
    }
    if (uhx - ulx + pad_w > uw) uw = uhx - ulx + pad_w;
    if (uhz - ulz + pad_h > uh) uh = uhz - ulz + pad_h;
  }
  if (uw > p.mw) uw = p.mw;
  if (uh > p.mh) uh = p.mh;
  rb.slab_stride = ((int)ceil(area) + 3) & ~3;
  rb.max_rows = (int)ceil(uh);
  rb.max_union = (((int)ceil(uw) + 3) & ~3) * rb.max_rows;
  const size_t lds = ((size_t)rb.slab_stride + 64) * 4 + kGeomBytes + (size_t)rb.max_rows * plan.P * 4;
  rb.fits = lds <= (size_t)kMaxLdsBytes;
}

const RigBound* rig_bound(const dm_params& p, const Plan& plan, const dm_frame& f0, int slack) {
  thread_local RigBound slots[4] = {};
  thread_local int next = 0;
  for (RigBound& r : slots)
    if (r.valid && r.slack == slack && r.P == plan.P && memcmp(&r.key, &p, sizeof(dm_params)) == 0 &&
        r.pitch[0] == f0.Rp[4] && r.pitch[1] == f0.Rp[5] && r.pitch[2] == f0.Rp[7] && r.pitch[3] == f0.Rp[8])
      return &r;
  RigBound& r = slots[next];
  next = (next + 1) % 4;
  r = RigBound{};
  compute_bound(p, plan, f0, slack, r);
  return &r;
}

using StripKernel = void (*)(StripArgs);

StripKernel pick_strip_kernel(bool is_max, bool has_valid, bool has_value, bool lean) {
#define DM_S(M) {{{k_strip_scatter<M, false, false, false>, k_strip_scatter<M, false, false, true>},   \
                  {k_strip_scatter<M, false, true, false>, k_strip_scatter<M, false, true, true>}},    \
                 {{k_strip_scatter<M, true, false, false>, k_strip_scatter<M, true, false, false>},    \
                  {k_strip_scatter<M, true, true, false>, k_strip_scatter<M, true, true, false>}}}
  // [min | max][has_valid][has_value][lean]   (lean implies no valid map)
  static const StripKernel table[2][2][2][2] = {DM_S(kMin), DM_S(kMax)};
#undef DM_S
  return table[is_max ? 1 : 0][has_valid][has_value][lean && !has_valid];
}

struct Layout {               // workspace of the strip path
  float* frames;              // (B, 32)
  Win16* g_wins;              // (B, kMaxStrips)
  Win16* g_unions;            // (B)
  int* status;
  float* sink;                // kSinks x 64 B
  uint32_t* g_covers;         // (B, max_rows, P)
  float* slabs;
  size_t slab_bytes;
};

size_t tables_bytes(int B, int rows, int P) {
  return up256((size_t)B * sizeof(dm_frame)) + up256((size_t)B * strip::kMaxStrips * sizeof(Win16)) +
         up256((size_t)B * sizeof(Win16)) + 256 + kSinks * 64 + up256((size_t)B * rows * P * 4);
}

bool carve(void* ws, size_t ws_bytes, int B, int rows, int P, Layout& l) {
  if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return false;
  const size_t t = tables_bytes(B, rows, P);
  if (ws_bytes < t) return false;
  unsigned char* base = static_cast<unsigned char*>(ws);
  l.frames = reinterpret_cast<float*>(base); base += up256((size_t)B * sizeof(dm_frame));
  l.g_wins = reinterpret_cast<Win16*>(base); base += up256((size_t)B * strip::kMaxStrips * sizeof(Win16));
  l.g_unions = reinterpret_cast<Win16*>(base); base += up256((size_t)B * sizeof(Win16));
  l.status = reinterpret_cast<int*>(base); base += 256;
  l.sink = reinterpret_cast<float*>(base); base += kSinks * 64;
  l.g_covers = reinterpret_cast<uint32_t*>(base); base += up256((size_t)B * rows * P * 4);
  l.slabs = reinterpret_cast<float*>(base);
  l.slab_bytes = ws_bytes - t;
  return true;
}

template <class K, class... Args>
inline hipError_t launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, const Args&... args) {
  hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
  return hipGetLastError();
}

hipError_t raise_lds_limit(const void* key) {
  static thread_local const void* done[32][8] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 8)
    for (int i = 0; i < 32; ++i)
      if (done[i][dev] == key) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 8)
    for (int i = 0; i < 32; ++i)
      if (!done[i][dev]) { done[i][dev] = key; break; }
  return hipSuccess;
}

// One pass over the channels of `out`: scatter (+ owned groups straight to the map) and merge.
hipError_t strip_pass(const dm_params& p, const Plan& plan, const RigBound& rb, const Layout& l,
                      const float* depth, const float* value, const uint8_t* valid, float* out,
                      uint8_t* mask, int oc_total, float fill, bool is_max, size_t slab_bytes,
                      hipStream_t s) {
  StripArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.W = p.W; sa.H = p.H;
  sa.clip = p.clip_border > 0 ? p.clip_border : 0;
  sa.flip_h = p.flip_h != 0; sa.to_global = p.to_global != 0;
  sa.cx = p.cx; sa.cy = p.cy; sa.fx = p.fx; sa.fy = p.fy; sa.res = p.res;
  sa.res_inv = plan.res_inv; sa.fx_inv = plan.fx_inv; sa.fy_inv = plan.fy_inv;
  sa.dmin = p.dmin; sa.dmax = p.dmax;
  sa.hmax = p.has_hmax ? p.hmax : INFINITY;
  sa.Hm1 = (float)(p.H - 1); sa.mhm1 = (float)(p.mh - 1);
  sa.wp = plan.wp; sa.P = plan.P;
  sa.dc = p.dc; sa.valid_c = p.valid_c;
  sa.oc_total = oc_total;
  sa.slab_stride = rb.slab_stride;
  sa.table_off = rb.slab_stride + 64;
  sa.max_rows = rb.max_rows;
  sa.fill = fill;
  sa.b0 = 0;
  sa.frames = l.frames;
  sa.depth = depth; sa.value = value; sa.valid = valid;
  sa.slabs = l.slabs;
  sa.out = out; sa.mask = mask; sa.mh = p.mh; sa.mw = p.mw;
  sa.g_wins = l.g_wins; sa.g_unions = l.g_unions; sa.g_covers = l.g_covers; sa.status = l.status; sa.sink = l.sink;
  sa.cfg = plan.cfg;
  const bool has_valid = valid != nullptr, has_value = value != nullptr;
  const StripKernel kfn = pick_strip_kernel(is_max, has_valid, has_value, plan.lean);
  hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn));
  if (e != hipSuccess) return e;
  const size_t lds_bytes = ((size_t)rb.slab_stride + 64) * 4 + kGeomBytes + (size_t)rb.max_rows * plan.P * 4;
  // channel groups: the slabs of one group fit the slab region
  const size_t per_channel = (size_t)p.B * plan.P * rb.slab_stride * 4;
  int group = (int)(slab_bytes / (per_channel ? per_channel : 1));
  if (group < 1) return hipErrorNotSupported;
  if (group > oc_total) group = oc_total;
  if (group > 65535) group = 65535;
  const int merge_blocks = (rb.max_union / 4 + kMergeThreads - 1) / kMergeThreads;
  for (int ch0 = 0; ch0 < oc_total; ch0 += group) {
    const int oc = oc_total - ch0 < group ? oc_total - ch0 : group;
    sa.oc = oc; sa.ch0 = ch0;
    e = launch(kfn, dim3(plan.P, oc, p.B), dim3(kScatterThreads), lds_bytes, s, sa);
    if (e != hipSuccess) return e;
    StripMergeArgs ma;
    ma.b0 = 0; ma.oc = oc; ma.ch0 = ch0; ma.oc_total = oc_total; ma.mh = p.mh; ma.mw = p.mw;
    ma.P = plan.P; ma.slab_stride = rb.slab_stride; ma.max_rows = rb.max_rows; ma.fill = fill;
    ma.g_wins = l.g_wins; ma.g_unions = l.g_unions; ma.g_covers = l.g_covers;
    ma.slabs = l.slabs; ma.out = out; ma.mask = mask;
    // grid.y = frames * channels <= 65535 per launch
    const int per_launch = 65535 / oc > 0 ? 65535 / oc : 1;
    for (int b0 = 0; b0 < p.B; b0 += per_launch) {
      const int nb = p.B - b0 < per_launch ? p.B - b0 : per_launch;
      ma.b0 = b0;
      const dim3 g((unsigned)merge_blocks, (unsigned)(nb * oc));
      e = is_max ? launch(k_strip_merge<kMax>, g, dim3(kMergeThreads), 0, s, ma)
                 : launch(k_strip_merge<kMin>, g, dim3(kMergeThreads), 0, s, ma);
      if (e != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

thread_local int g_force_legacy = 0;       // dm_debug_force_legacy_window

}  // namespace

size_t strip_workspace_extra(const dm_params& p) {
  Plan plan;
  if (!make_plan(p, plan)) return 0;
  return tables_bytes(p.B, p.mh, strip::kMaxStrips);
}

// hipErrorNotSupported: the strip path does not apply to this call (nothing enqueued).
hipError_t run_strip(const dm_params& p, const dm_frame* frames_host, const float* depth,
                     const float* value, const uint8_t* valid, float* out, uint8_t* mask,
                     float* height, float* fused, uint8_t* fused_mask, void* ws, size_t ws_bytes,
                     hipEvent_t before_projection, hipEvent_t after_projection, hipStream_t s) {
  if (g_force_legacy || p.B > 65535) return hipErrorNotSupported;
  const int oc_total = p.vc ? p.vc : p.dc;
  if (reinterpret_cast<uintptr_t>(out) % 16 != 0 || reinterpret_cast<uintptr_t>(mask) % 4 != 0 ||
      reinterpret_cast<uintptr_t>(fused) % 16 != 0 || reinterpret_cast<uintptr_t>(fused_mask) % 4 != 0 ||
      reinterpret_cast<uintptr_t>(height) % 16 != 0 || reinterpret_cast<uintptr_t>(depth) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(value) % 16 != 0)
    return hipErrorNotSupported;
  thread_local dm_params plan_key = {};
  thread_local Plan plan;
  thread_local bool plan_ok = false, plan_valid = false;
  thread_local int plan_forced = 0;
  if (!plan_valid || plan_forced != g_force_strips || memcmp(&plan_key, &p, sizeof(dm_params)) != 0) {
    plan_key = p;
    plan_forced = g_force_strips;
    plan_ok = make_plan(p, plan);
    plan_valid = true;
  }
  if (!plan_ok) return hipErrorNotSupported;
  const int slack = validate_frames(p, plan.cfg, frames_host, p.B);
  if (slack < 0) return hipErrorNotSupported;
  const RigBound* rb = rig_bound(p, plan, frames_host[0], slack);
  if (!rb->fits) return hipErrorNotSupported;
  Layout l;
  if (!carve(ws, ws_bytes, p.B, rb->max_rows, plan.P, l)) return hipErrorNotSupported;
  const size_t hm = (height && value) ? up256((size_t)p.B * p.dc * p.mh * p.mw) : 0;
  if (l.slab_bytes < hm + (size_t)p.B * plan.P * rb->slab_stride * 4) return hipErrorNotSupported;

  hipError_t e = hipSuccess;
  if (before_projection) {
    e = hipEventRecord(before_projection, s);
    if (e != hipSuccess) return e;
  }
  e = hipMemcpyAsync(l.frames, frames_host, (size_t)p.B * sizeof(dm_frame), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  const bool is_max = p.reduction == DM_REDUCE_MAX;
  e = strip_pass(p, plan, *rb, l, depth, value, valid, out, mask, oc_total, p.fill, is_max,
                 l.slab_bytes - hm, s);
  if (e != hipSuccess) return e;
  if (height && value) {      // maps.py:332-350: second projection of the heights, NINF fill, max
    uint8_t* scratch_mask = static_cast<unsigned char*>(ws) + ws_bytes - hm;
    e = strip_pass(p, plan, *rb, l, depth, nullptr, valid, height, scratch_mask, p.dc, -INFINITY, true,
                   l.slab_bytes - hm, s);
    if (e != hipSuccess) return e;
  }
  if (after_projection) {
    e = hipEventRecord(after_projection, s);
    if (e != hipSuccess) return e;
  }
  if (fused) {
    FuseArgs fa;
    fa.B = p.B; fa.b0 = 0; fa.accumulate = 0;
    fa.dc = oc_total; fa.mh = p.mh; fa.mw = p.mw; fa.fill = p.fill;
    fa.unions = l.g_unions; fa.maps = out; fa.fused = fused; fa.fused_mask = fused_mask;
    const dim3 g((unsigned)(((size_t)p.mh * p.mw / 4 + kFuseGroups - 1) / kFuseGroups), oc_total);
    const dim3 blk(kFuseGroups * kFuseLanes);
    e = is_max ? launch(k_fuse_unions<true>, g, blk, 0, s, fa)
               : launch(k_fuse_unions<false>, g, blk, 0, s, fa);
    if (e != hipSuccess) return e;
  }
  note_split(plan.P, 1, 1, 2);
  return hipSuccess;
}

}  // namespace dm

extern "C" __attribute__((visibility("default"))) int dm_debug_force_strips(int strips) {
  const int old = dm::g_force_strips;
  dm::g_force_strips = strips > 0 && strips <= dm::strip::kMaxStrips ? strips : 0;
  return old;
}

extern "C" __attribute__((visibility("default"))) int dm_debug_force_legacy_window(int on) {
  const int old = dm::g_force_legacy;
  dm::g_force_legacy = on != 0;
  return old;
}

// Host only (no GPU needed): the strip path's geometry for `p` and the given frames, exactly as
// the kernels derive it.  out_geom (B, 4 + 4 * kMaxStrips) int32: {ok, P, strip width, 0},
// union window {x0, z0, w, h}... see include/dungeon_maps_amd.h.  Returns the number of strips,
// 0 when the strip path does not apply to `p`, negative on bad arguments.
extern "C" __attribute__((visibility("default"))) int dm_debug_strip_geometry(
    const dm_params* p, const dm_frame* frames, int32_t* out_geom, uint32_t* out_covers,
    int32_t* out_bound) {
  using namespace dm;
  if (!p || !frames || !out_geom || p->B < 1) return -1;
  Plan plan;
  if (!make_plan(*p, plan)) return 0;
  const int slack = validate_frames(*p, plan.cfg, frames, p->B);
  if (out_bound) {
    out_bound[0] = slack; out_bound[1] = out_bound[2] = out_bound[3] = out_bound[4] = 0;
    if (slack >= 0) {
      RigBound rb = {};
      compute_bound(*p, plan, frames[0], slack, rb);
      out_bound[1] = rb.fits; out_bound[2] = rb.slab_stride; out_bound[3] = rb.max_rows;
      out_bound[4] = rb.max_union;
    }
  }
  const int stride = 8 + 4 * strip::kMaxStrips;
  for (int b = 0; b < p->B; ++b) {
    strip::FrameGeom g;
    strip::frame_geometry(plan.cfg, frames[b].Rp, g);
    int32_t* o = out_geom + (size_t)b * stride;
    o[0] = g.ok; o[1] = plan.P; o[2] = plan.wp; o[3] = slack;
    o[4] = g.U.x0; o[5] = g.U.z0; o[6] = g.U.w; o[7] = g.U.h;
    for (int s = 0; s < strip::kMaxStrips; ++s) {
      o[8 + 4 * s] = g.win[s].x0; o[9 + 4 * s] = g.win[s].z0;
      o[10 + 4 * s] = g.win[s].w; o[11 + 4 * s] = g.win[s].h;
    }
    if (out_covers)
      for (int z = 0; z < p->mh; ++z)
        for (int s = 0; s < plan.P; ++s)
          out_covers[((size_t)b * p->mh + z) * plan.P + s] = strip::row_cover(g.win[s], g.L[s], g.R[s], z);
  }
  return plan.P;
}

// GPU: the same geometry from k_strip_geometry_dump (the device's lane-parallel evaluation),
// for the test that host and device agree bit for bit.  frames_dev (B, 32) f32, geom_dev
// B * sizeof(FrameGeom) bytes of device scratch; copies back windows + union + ok as above.
extern "C" __attribute__((visibility("default"))) int dm_debug_strip_geometry_dev(
    const dm_params* p, const float* frames_dev, void* geom_dev, size_t geom_bytes, void* stream) {
  using namespace dm;
  if (!p || !frames_dev || !geom_dev || p->B < 1) return -1;
  if (geom_bytes < (size_t)p->B * sizeof(strip::FrameGeom)) return -(int)sizeof(strip::FrameGeom);
  Plan plan;
  if (!make_plan(*p, plan)) return 0;
  hipLaunchKernelGGL(k_strip_geometry_dump, dim3(p->B), dim3(64), 0, static_cast<hipStream_t>(stream),
                     plan.cfg, frames_dev, static_cast<strip::FrameGeom*>(geom_dev));
  return hipGetLastError() == hipSuccess ? plan.P : -2;
}
