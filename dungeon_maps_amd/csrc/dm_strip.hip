// Strip path of orth_project (max / min): host side.  See dm_strip_geometry.hpp for the
// geometry and dm_strip_kernels.hpp for the kernels.
//
// What the host does per call is pose independent: it validates the frame records (the
// rotations have the axis-aligned pattern the fast arithmetic needs, one camera pitch for the
// whole batch, magnitudes within the float32 slack) and launches k_strip_scatter +
// k_strip_combine with sizes taken from a bound that holds for EVERY yaw and position of the
// camera (cached per camera rig).  The frames' camera state travels in the scatter kernel's
// arguments (48 bytes per frame, up to 64 frames per launch): no copy and no table kernel in
// front of it -- the workgroups derive their frames' geometry themselves.
#include <math.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "dm_kernels.hpp"
#include "dm_strip_kernels.hpp"
#include "dm_strip_fused_kernels.hpp"

constexpr int kCombineEntriesMany = 4;      // list entries per thread of the combine kernel for value maps of many channels

namespace dm {

void note_split(int pc, int pr, int pd, int path);     // dm_window.hip (dm_debug_last_split)

namespace {

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

// Pose-independent plan of a call (the parameters alone).
struct Plan {
  int P, wp;
  float res_inv, fx_inv, fy_inv;
  bool lean;
  int live[strip::kMaxStrips];
  float ax_lo[strip::kMaxStrips], ax_hi[strip::kMaxStrips];   // ray slopes of each strip's live columns
  float ay_lo, ay_hi;                                         // ... and of the live rows
};

// The plan plus the batch's camera pitch: the device-side Cfg, and launch sizes that hold for
// every camera yaw / position.
struct Rig {
  dm_params key;
  float pitch[4];               // Rp[4], Rp[5], Rp[7], Rp[8]
  int slack;                    // cells of slack the bound was computed for
  int P;
  int wp;                       // strip width the cones were derived for
  bool valid, fits;
  strip::Cfg cfg;
  strip::RigArgs args;          // cfg as the kernels take it
  int slab_stride;              // cells of the largest window any strip can have
  int max_rows;                 // rows of the largest union window
  int max_union;                // cells of the largest union window
};

thread_local int g_force_strips = 0;       // dm_debug_force_strips

bool make_plan(const dm_params& p, Plan& plan, int strips = 0) {
  if (p.reduction != DM_REDUCE_MAX && p.reduction != DM_REDUCE_MIN) return false;
  if (p.mw % 4 != 0 || p.W % 4 != 0) return false;
  if (p.mw > 32767 || p.mh > 32767 || (int64_t)p.mh * p.mw >= (1ll << 28)) return false;   // (32-bit byte offsets)
  if ((int64_t)p.H * p.W >= (1ll << 28) || p.H >= (1 << 23) || p.W >= (1 << 23)) return false;   // (the same for the images; 24-bit row multiplies)
  if (!(p.fill == p.fill)) return false;
  if (!p.has_dmin || !p.has_dmax || !(p.dmin >= 0.0f) || !(p.dmax >= p.dmin) || !isfinite(p.dmax))
    return false;
  Parts parts = choose_parts(p, 1, 1);
  if (g_force_strips > 0) strips = g_force_strips;   // tests: whatever the cost model says
  if (strips > 0) {              // this many column strips (more than the model's: narrower windows)
    parts.pc = strips; parts.pr = 1; parts.pd = 1;
    parts.wp = ((p.W + parts.pc - 1) / parts.pc + 3) & ~3;
    const int wp32 = (parts.wp + 31) & ~31;     // whole 128-byte lines where that keeps every strip
    if (p.W % 32 == 0 && (parts.pc - 1) * wp32 < p.W) parts.wp = wp32;
    parts.hp = p.H;
  }
  if (parts.pr != 1 || parts.pd != 1 || parts.pc > strip::kMaxStrips || parts.wp % 4 != 0) return false;
  if (!exact_reciprocal(p.res, &plan.res_inv) || !exact_reciprocal(p.fx, &plan.fx_inv) ||
      !exact_reciprocal(p.fy, &plan.fy_inv) || p.res < 1e-6f || p.res > 1e6f || p.fx < 1e-6f ||
      p.fx > 1e6f || p.fy < 1e-6f || p.fy > 1e6f)
    return false;
  plan.P = parts.pc;
  plan.wp = parts.wp;
  const int clip = p.clip_border > 0 ? p.clip_border : 0;
  const int r0 = clip, r1 = p.H - clip;
  if (r0 >= r1) return false;
  double ay[2];
  const int rs[2] = {r0, r1 - 1};
  for (int i = 0; i < 2; ++i) {
    double yr = rs[i];
    if (p.flip_h) yr = (double)(p.H - 1) - yr;
    ay[i] = (yr - (double)p.cy) / (double)p.fy;
  }
  plan.ay_lo = (float)(ay[0] < ay[1] ? ay[0] : ay[1]);
  plan.ay_hi = (float)(ay[0] < ay[1] ? ay[1] : ay[0]);
  bool any = false;
  for (int s = 0; s < strip::kMaxStrips; ++s) {
    plan.live[s] = 0; plan.ax_lo[s] = plan.ax_hi[s] = 0.0f;
    if (s >= plan.P) continue;
    int q0 = s * plan.wp, q1 = q0 + plan.wp < p.W ? q0 + plan.wp : p.W;
    if (q0 < clip) q0 = clip;
    if (q1 > p.W - clip) q1 = p.W - clip;
    plan.live[s] = q0 < q1;
    if (!plan.live[s]) continue;
    any = true;
    plan.ax_lo[s] = (float)(((double)q0 - (double)p.cx) / (double)p.fx);
    plan.ax_hi[s] = (float)(((double)(q1 - 1) - (double)p.cx) / (double)p.fx);
  }
  if (!any) return false;
  plan.lean = !p.valid_c && isfinite(p.dmin) && !p.has_hmax && p.clip_border <= 0;
  return true;
}

// The frame records the strip path can take: axis-aligned rotations (the fast arithmetic) whose
// yaw block is orthonormal (the launch bound rotates the cones rigidly), one pitch for the
// batch, finite values; returns the cells of slack that cover every frame (>= what the device
// computes for it) or -1.
int validate_frames(const dm_params& p, const dm_frame* f, int B) {
  double worst = 0.0;
  const double inv = 1.0 / (double)p.res;
  for (int b = 0; b < B; ++b) {
    const float* rp = f[b].Rp;
    const float* ry = f[b].Ry;
    if (!(rp[0] == 1.0f && rp[1] == 0.0f && rp[2] == 0.0f && rp[3] == 0.0f && rp[6] == 0.0f)) return -1;
    if (p.to_global) {
      if (!(ry[1] == 0.0f && ry[3] == 0.0f && ry[4] == 1.0f && ry[5] == 0.0f && ry[7] == 0.0f)) return -1;
      // (a scaled or sheared yaw block would move points further than the bound assumes)
      const double y0 = ry[0], y2 = ry[2], y6 = ry[6], y8 = ry[8];
      if (!(fabs(y0 * y0 + y2 * y2 - 1.0) < 1e-5 && fabs(y6 * y6 + y8 * y8 - 1.0) < 1e-5 &&
            fabs(y0 * y6 + y2 * y8) < 1e-5))
        return -1;
    }
    if (rp[4] != f[0].Rp[4] || rp[5] != f[0].Rp[5] || rp[7] != f[0].Rp[7] || rp[8] != f[0].Rp[8]) return -1;
    if (!isfinite(f[b].cam_height)) return -1;
    double m = fabs((double)f[b].width_offset) + fabs((double)f[b].height_offset) + (double)p.mh;
    if (p.to_global) m += 2.0 * (fabs((double)f[b].tx) + fabs((double)f[b].tz)) * inv;
    m = 2.0 * m;
    if (!(m == m) || !isfinite(m)) return -1;
    if (m > worst) worst = m;
  }
  return worst > 1e9 ? -1 : (int)worst;       // (magnitude; the rig adds its reach)
}

// Cfg and launch bound of a rig.  The bound: the truncated cones are rotated in 1440 steps;
// between two steps an extent grows by at most Rmax * dtheta.
void compute_rig(const dm_params& p, const Plan& plan, const float* pitch4, int magnitude, Rig& rg) {
  rg.key = p; rg.P = plan.P; rg.wp = plan.wp; rg.valid = true; rg.fits = false;
  memcpy(rg.pitch, pitch4, sizeof(rg.pitch));
  float Rp[9] = {1.0f, 0.0f, 0.0f, 0.0f, pitch4[0], pitch4[1], 0.0f, pitch4[2], pitch4[3]};
  strip::Cfg& c = rg.cfg;
  memset(&c, 0, sizeof(c));
  c.P = plan.P; c.mw = p.mw; c.mh = p.mh; c.flip_h = p.flip_h != 0;
  c.inv = (float)(1.0 / (double)p.res);
  memcpy(c.live, plan.live, sizeof(c.live));
  strip::cfg_rig(c, Rp, plan.ax_lo, plan.ax_hi, plan.ay_lo, plan.ay_hi, p.dmin, p.dmax);
  rg.args = strip::rig_args(c, Rp, plan.ax_lo, plan.ax_hi, plan.ay_lo, plan.ay_hi, p.dmin, p.dmax);
  // slack that covers every frame of the batch (>= the device's own: same formula, larger m)
  const double slack_d = 2.0 + 16.0 * ((double)magnitude + 2.0 * (double)c.reach) * (1.0 / 8388608.0) + 0.01;
  rg.slack = slack_d > 16.0 ? -1 : (int)ceil(slack_d);
  if (!c.cone_ok || rg.slack < 0) return;
  double rmax = 0.0;
  for (int s = 0; s < plan.P; ++s)
    for (int k = 0; k < 8; ++k) {
      const double r = sqrt((double)c.cxl[s][k] * c.cxl[s][k] + (double)c.czl[s][k] * c.czl[s][k]);
      if (r > rmax) rmax = r;
    }
  if (!isfinite(rmax)) return;
  const int steps = 1440;
  const double dtheta = 2.0 * M_PI / steps;
  const double lip = rmax * dtheta;
  const double pad_w = lip + 2.0 * rg.slack + 10.0, pad_h = lip + 2.0 * rg.slack + 4.0;
  double area = 0.0, uw = 0.0, uh = 0.0;
  for (int i = 0; i < steps; ++i) {
    const double cs = cos(i * dtheta), sn = sin(i * dtheta);
    double ulx = INFINITY, uhx = -INFINITY, ulz = INFINITY, uhz = -INFINITY;
    for (int s = 0; s < plan.P; ++s) {
      if (!c.live[s]) continue;
      double lx = INFINITY, hx = -INFINITY, lz = INFINITY, hz = -INFINITY;
      for (int k = 0; k < 8; ++k) {
        const double x = cs * c.cxl[s][k] + sn * c.czl[s][k], z = -sn * c.cxl[s][k] + cs * c.czl[s][k];
        lx = x < lx ? x : lx; hx = x > hx ? x : hx; lz = z < lz ? z : lz; hz = z > hz ? z : hz;
      }
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
  rg.slab_stride = ((int)ceil(area) + 3) & ~3;
  rg.max_rows = ((int)ceil(uh) + 3) & ~3;       // (whole float4s: the LDS tables behind it stay 16-byte aligned)
  const int uw4 = ((int)ceil(uw) + 2 * strip::kSpanAlign + 3) & ~3;
  rg.max_union = uw4 * rg.max_rows;
  // (the shared-group entries hold 12 bits of row and of float4 group: the device checks the same)
  rg.fits = strip_lds_bytes(rg.slab_stride, rg.max_rows, p.H, plan.P) <= (size_t)kMaxLdsBytes &&
            rg.max_rows <= kListMaxRows && uw4 <= 4 * kListMaxGroups;
}

// The magnitude only matters through the slack it implies, so it is quantised: one rig serves
// every batch whose frames stay within the same 2^17 cells (below 2^18: one rig for all), and a
// rig is a pure function of (parameters, plan, pitch, quantised magnitude) -- what a prepared
// batch's plan records is enough to find the same rig again.
// Idempotent (a prepared batch's plan records the quantised value and rig_of quantises what it is
// given): the next multiple of 2^17 at or above the magnitude, at least 2^18.
inline int quantised_magnitude(int magnitude) {
  if (magnitude <= (1 << 18)) return 1 << 18;
  const int64_t up = (((int64_t)magnitude + (1 << 17) - 1) >> 17) << 17;
  return up > 0x7ffe0000 ? 0x7ffe0000 : (int)up;       // (the largest multiple of 2^17 an int holds)
}

const Rig* rig_of(const dm_params& p, const Plan& plan, const float* pitch4, int magnitude) {
  thread_local Rig slots[4] = {};
  thread_local int slot_mag[4] = {0, 0, 0, 0};
  thread_local int next = 0;
  const int mag_q = quantised_magnitude(magnitude);
  for (int i = 0; i < 4; ++i) {
    Rig& r = slots[i];
    if (r.valid && r.P == plan.P && r.wp == plan.wp && slot_mag[i] == mag_q &&
        memcmp(&r.key, &p, sizeof(dm_params)) == 0 && memcmp(r.pitch, pitch4, sizeof(r.pitch)) == 0)
      return &r;
  }
  Rig& r = slots[next];
  slot_mag[next] = mag_q;
  next = (next + 1) % 4;
  r = Rig{};
  compute_rig(p, plan, pitch4, mag_q, r);
  return &r;
}

using StripKernel = void (*)(StripArgs);

// Compact planes (k_strip_scatter<..., PLANES> + k_strip_combine_planes): calls of at most four strips and two
// output channels.  dm_debug_planes: -1 the default (on), 0 / 1.
thread_local int g_planes = -1;
StripKernel pick_planes_kernel(bool is_max, bool has_valid, bool has_value, bool lean, bool nt_fill) {
#define DM_K(M, VALID, VALUE, LEAN) {k_strip_scatter<M, VALID, VALUE, LEAN, kProject, false, true>, k_strip_scatter<M, VALID, VALUE, LEAN, kProject, true, true>}
#define DM_S(M) {{{DM_K(M, false, false, false), DM_K(M, false, false, true)},   \
                  {DM_K(M, false, true, false), DM_K(M, false, true, true)}},    \
                 {{DM_K(M, true, false, false), DM_K(M, true, false, false)},    \
                  {DM_K(M, true, true, false), DM_K(M, true, true, false)}}}
  static const StripKernel table[2][2][2][2][2] = {DM_S(kMin), DM_S(kMax)};
#undef DM_S
#undef DM_K
  return table[is_max ? 1 : 0][has_valid][has_value][lean && !has_valid][nt_fill];
}

StripKernel pick_strip_kernel(bool is_max, bool has_valid, bool has_value, bool lean, bool nt_fill) {
#define DM_K(M, VALID, VALUE, LEAN) {k_strip_scatter<M, VALID, VALUE, LEAN, kProject, false>, k_strip_scatter<M, VALID, VALUE, LEAN, kProject, true>}
#define DM_S(M) {{{DM_K(M, false, false, false), DM_K(M, false, false, true)},   \
                  {DM_K(M, false, true, false), DM_K(M, false, true, true)}},    \
                 {{DM_K(M, true, false, false), DM_K(M, true, false, false)},    \
                  {DM_K(M, true, true, false), DM_K(M, true, true, false)}}}
  // [min | max][has_valid][has_value][lean][fill stores non-temporal]   (lean implies no valid map)
  static const StripKernel table[2][2][2][2][2] = {DM_S(kMin), DM_S(kMax)};
#undef DM_S
#undef DM_K
  return table[is_max ? 1 : 0][has_valid][has_value][lean && !has_valid][nt_fill];
}

// dm_debug_fill_split: where the fill duty of the map rows outside a frame's union window runs
// (StripArgs::defer_outer / head_share).  -1: under the pixel loop with the rest (rounds 2-3);
// 0..8: out of the loop -- that many of every eight such rows of a wave in the kernel's head, the
// others in the combine kernel.
constexpr int kFillSplitDefault = -1;     // (measured: no gain in the head, a loss in the combine kernel -- DESIGN 4.6)
thread_local int g_fill_split = kFillSplitDefault;

// Whether the call takes the kernel's STREAMING variant (template parameter NT_FILL): the fill value's
// stores bypass the caches (dm_pixel.hpp kFillCachePolicy) and the depth maps are loaded non-temporally
// (k_strip_scatter kDepthPolicy).
thread_local int g_force_nt_fill = -1;      // dm_debug_force_nt_fill: -1 the rule below, 0 never, 1 always
inline bool nt_fill_pays(const dm_params& p, int oc_total, bool fuse_follows) {
  if (g_force_nt_fill >= 0) return g_force_nt_fill != 0;
  // the streaming variant (fill stores and depth loads non-temporal): where the call's own batch fuse
  // follows, or where what the call reads and writes once -- depth maps + maps + masks -- is more than the
  // Infinity Cache could keep beside anything else (half of its 256 MiB)
  const size_t once = (size_t)p.B * p.dc * p.H * p.W * 4 + (size_t)p.B * oc_total * p.mh * p.mw * 5;
  return fuse_follows || once > ((size_t)128 << 20);
}


// Value maps of kListMinChannels channels or more (one depth channel for all of them): the cell of
// every pixel is computed once (index pass) and the channels scatter from that list (value pass).
constexpr int kListMinChannels = 3;
StripKernel pick_index_kernel(bool has_valid, bool lean) {
  if (lean && !has_valid) return k_strip_scatter<kMax, false, false, true, kIndexOut>;
  return has_valid ? k_strip_scatter<kMax, true, false, false, kIndexOut>
                   : k_strip_scatter<kMax, false, false, false, kIndexOut>;
}
// The value pass with the workgroups of one (frame, strip) on one XCD (StripArgs::xcd_units): that XCD's L2 then
// fetches the unit's part of the pixel list once (PMC at cfg3: 3.72 -> 3.36 GB fetched by the pass, the list read
// ~5.5 times per call instead of ~14) -- and the pass takes 2 % LONGER (1 442 / 1 451 against 1 416 / 1 421 us on one
// box): the re-reads were hits in the Infinity Cache, which cost HBM nothing, while a fixed 32 units per XCD
// balance worse than blocks dealt one by one.  Off.
constexpr bool g_no_xcd_units = true;

StripKernel pick_value_kernel(bool is_max) {
  return is_max ? k_strip_scatter<kMax, false, true, true, kFromList>
                : k_strip_scatter<kMin, false, true, true, kFromList>;
}
inline size_t pixel_list_bytes(const dm_params& p) {     // (B, P, H, wp) uint16: P * wp <= W + 32 per strip
  return up256((size_t)p.B * p.H * ((size_t)p.W + 32 * strip::kMaxStrips) * 2);
}
thread_local int g_no_value_list = 0;      // dm_debug_strip_value_list(0): every channel recomputes its cells
thread_local int g_last_info[4] = {0, 0, 0, 0};   // dm_debug_last_strip_info
thread_local size_t g_strip_slab_budget = 0;    // dm_debug_strip_slab_budget: bytes of slabs per channel group (0: all there is)
inline bool list_shape(const dm_params& p) { return p.vc >= kListMinChannels && p.dc == 1; }   // (sizes the workspace)
inline bool wants_pixel_list(const dm_params& p) { return list_shape(p) && !g_no_value_list; }

// Workspace of the strip path:
//   [frame tables: wins (B, 8), unions (B), list counts (B, 8) | shared-group lists (B, P, seg_cap) |
//    slabs ... | pixel list]
// The tables are written by the scatter kernel for the kernels behind it; nothing is staged
// from the host.  seg_cap = slab_cells / 4 (a strip lists at most the groups of its window).
inline size_t tables_bytes(const dm_params& p) {
  return up256((size_t)p.B * strip::kMaxStrips * sizeof(Win16)) + up256((size_t)p.B * sizeof(Win16)) +
         up256((size_t)p.B * strip::kMaxStrips * sizeof(int));
}
inline size_t lists_bytes(const dm_params& p, int P, int slab_cells) {
  return up256((size_t)p.B * P * (size_t)(slab_cells / 4) * sizeof(uint32_t));
}
// (from the parameters alone: as many strips and as large windows as LDS can hold)
inline size_t lists_bytes_bound(const dm_params& p) {
  return lists_bytes(p, strip::kMaxStrips, kMaxLdsBytes / 4);
}

struct Layout {
  FrameTables t;
  float* slabs;
  size_t slab_bytes;
  uint16_t* pixel_list;       // value maps of many channels: the pixels' cells (or NULL)
};

bool carve(const dm_params& p, int P, int slab_cells, void* ws, size_t ws_bytes, Layout& l) {
  if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return false;
  ws_bytes = ws_bytes / 256 * 256;
  const size_t lb = list_shape(p) ? pixel_list_bytes(p) : 0;
  const size_t head = tables_bytes(p) + lists_bytes(p, P, slab_cells);
  if (ws_bytes < head + 256 + lb) return false;
  unsigned char* base = static_cast<unsigned char*>(ws);
  l.t.wins = reinterpret_cast<Win16*>(base); base += up256((size_t)p.B * strip::kMaxStrips * sizeof(Win16));
  l.t.unions = reinterpret_cast<Win16*>(base); base += up256((size_t)p.B * sizeof(Win16));
  l.t.counts = reinterpret_cast<int*>(base); base += up256((size_t)p.B * strip::kMaxStrips * sizeof(int));
  l.t.list = reinterpret_cast<uint32_t*>(base); base += lists_bytes(p, P, slab_cells);
  l.slabs = reinterpret_cast<float*>(base);
  l.slab_bytes = ws_bytes - head - lb;
  l.pixel_list = lb ? reinterpret_cast<uint16_t*>(static_cast<unsigned char*>(ws) + ws_bytes - lb) : nullptr;
  return true;
}

template <class K, class... Args>
inline hipError_t launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s, const Args&... args) {
  hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
  return hipGetLastError();
}

hipError_t raise_lds_limit(const void* key) {
  static thread_local const void* done[128][8] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 8)
    for (int i = 0; i < 128; ++i)
      if (done[i][dev] == key) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes);
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 8)
    for (int i = 0; i < 128; ++i)
      if (!done[i][dev]) { done[i][dev] = key; break; }
  return hipSuccess;
}

// Where a launch's camera state comes from: the host's frame records (copied into the kernel
// arguments, up to kPoseFrames frames per launch) or a device buffer of StripPose records.
struct PoseSource {
  const dm_frame* host;       // or NULL
  const float* dev;           // (B, kPoseFloats) or NULL
  bool to_global;
};

inline void fill_poses(StripArgs& sa, const PoseSource& src, int b0, int nb) {
  sa.b0 = b0;
  sa.poses_dev = src.dev;
  if (src.dev) return;
  for (int i = 0; i < nb; ++i) {
    const dm_frame& f = src.host[b0 + i];
    StripPose& q = sa.poses[i];
    // (a local map: neutral yaw, no translation -- exact: x * 1 + z * 0 + 0)
    q.y0 = src.to_global ? f.Ry[0] : 1.0f; q.y2 = src.to_global ? f.Ry[2] : 0.0f;
    q.y6 = src.to_global ? f.Ry[6] : 0.0f; q.y8 = src.to_global ? f.Ry[8] : 1.0f;
    q.tx = src.to_global ? f.tx : 0.0f; q.tz = src.to_global ? f.tz : 0.0f;
    q.wo = f.width_offset; q.ho = f.height_offset; q.cam_h = f.cam_height;
    q.pad0 = q.pad1 = q.pad2 = 0.0f;
  }
}

// One pass over the channels of `out`: scatter (+ owned groups straight to the map) and combine,
// kPoseFrames frames per launch.
hipError_t strip_pass(const dm_params& p, const Plan& plan, const Rig& rg, const PoseSource& poses,
                      const Layout& l, const float* depth, const float* value, const uint8_t* valid,
                      float* out, uint8_t* mask, int oc_total, float fill, bool is_max, size_t slab_bytes,
                      bool fuse_follows, int* status, hipStream_t s) {
  thread_local StripArgs sa;       // (3.5 KB: not on the stack of every caller)
  memset(&sa, 0, offsetof(StripArgs, poses));
  sa.W = p.W; sa.H = p.H;
  sa.clip = p.clip_border > 0 ? p.clip_border : 0;
  sa.flip_h = p.flip_h != 0;
  sa.cx = p.cx; sa.cy = p.cy; sa.fx = p.fx; sa.fy = p.fy; sa.res = p.res;
  sa.res_inv = plan.res_inv; sa.fx_inv = plan.fx_inv; sa.fy_inv = plan.fy_inv;
  sa.dmin = p.dmin; sa.dmax = p.dmax;
  sa.hmax = p.has_hmax ? p.hmax : INFINITY;
  sa.Hm1 = (float)(p.H - 1); sa.mhm1 = (float)(p.mh - 1);
  sa.p4 = rg.pitch[0]; sa.p5 = rg.pitch[1]; sa.p7 = rg.pitch[2]; sa.p8 = rg.pitch[3];
  sa.wp = plan.wp; sa.P = plan.P;
  sa.dc = p.dc; sa.valid_c = p.valid_c;
  sa.oc_total = oc_total;
  sa.slab_stride = rg.slab_stride;
  sa.max_rows = rg.max_rows;
  sa.seg_cap = rg.slab_stride / 4;
  sa.fill = fill;
  sa.depth = depth; sa.value = value; sa.valid = valid;
  sa.slabs = l.slabs;
  sa.out = out; sa.mask = mask; sa.mh = p.mh; sa.mw = p.mw;
  sa.t = l.t;
  sa.status = status;
  // (height maps and value maps of few channels: the launch pairs whose combine kernel is
  // k_strip_combine_one, which takes its share of the fill duty)
  sa.defer_outer = g_fill_split >= 0 && oc_total < kListMinChannels;
  sa.head_share = sa.defer_outer ? g_fill_split : 0;
  sa.xcd_units = 0;
  sa.rig = rg.args;
#ifdef DM_STAMPS
  sa.stamps = g_stamp_buffer;
#endif
  const bool has_valid = valid != nullptr, has_value = value != nullptr;
  const bool from_list = has_value && l.pixel_list != nullptr && wants_pixel_list(p);
  sa.list = from_list ? l.pixel_list : nullptr;
  // Compact planes: the shared groups of a frame numbered frame-wide, P planes of plane_cap float4 per (frame,
  // channel) in the slab region, the groups' places in the first plane_cap words of the frame's list segments.
  // plane_cap: a frame cannot share more groups than half of what its P windows hold.
  const int plane_cap = ((plan.P * (rg.slab_stride / 4) / 2 + 255) / 256) * 256;
  const bool planes = !from_list && oc_total <= 2 && plan.P <= 4 && g_planes != 0 && !sa.defer_outer &&
                      (size_t)p.B * oc_total * plan.P * plane_cap * 16 <= slab_bytes &&
                      plane_cap <= plan.P * sa.seg_cap;
  sa.plane_cap = planes ? plane_cap : 0;
  const bool nt = nt_fill_pays(p, oc_total, fuse_follows);
  const StripKernel kfn = from_list ? pick_value_kernel(is_max)
                          : planes ? pick_planes_kernel(is_max, has_valid, has_value, plan.lean, nt)
                                   : pick_strip_kernel(is_max, has_valid, has_value, plan.lean, nt);
  hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn));
  if (e != hipSuccess) return e;
  const size_t lds_bytes = strip_lds_bytes(rg.slab_stride, rg.max_rows, p.H, plan.P);
  if (lds_bytes > (size_t)kMaxLdsBytes) return hipErrorNotSupported;
  StripKernel ifn = nullptr;
  if (from_list) {            // the index pass: every pixel's cell inside its strip's window, once for all channels
    ifn = pick_index_kernel(has_valid, plan.lean);
    e = raise_lds_limit(reinterpret_cast<const void*>(ifn));
    if (e != hipSuccess) return e;
  }
  // channel groups: the slabs of one group fit the slab region
  const size_t per_channel = (size_t)p.B * plan.P * rg.slab_stride * 4;
  if (g_strip_slab_budget && slab_bytes > g_strip_slab_budget) slab_bytes = g_strip_slab_budget;
  int group = (int)(slab_bytes / (per_channel ? per_channel : 1));
  if (group < 1) return hipErrorNotSupported;
  if (group > oc_total) group = oc_total;
  if (group > 65535) group = 65535;
  for (int b0 = 0; b0 < p.B; b0 += kPoseFrames) {
    const int nb = p.B - b0 < kPoseFrames ? p.B - b0 : kPoseFrames;
    fill_poses(sa, poses, b0, nb);
    if (from_list) {
      StripArgs& ia = sa;       // (the same arguments, no maps, one "channel")
      const float* keep_value = ia.value; float* keep_out = ia.out; uint8_t* keep_mask = ia.mask;
      ia.value = nullptr; ia.out = nullptr; ia.mask = nullptr; ia.oc = 1; ia.ch0 = 0; ia.oc_total = 1;
      e = launch(ifn, dim3(plan.P, 1, nb), dim3(kScatterThreads), lds_bytes, s, ia);
      ++g_last_info[0];
      ia.value = keep_value; ia.out = keep_out; ia.mask = keep_mask; ia.oc_total = oc_total;
      if (e != hipSuccess) return e;
    }
    for (int ch0 = 0; ch0 < oc_total; ch0 += group) {
      const int oc = oc_total - ch0 < group ? oc_total - ch0 : group;
      sa.oc = oc; sa.ch0 = ch0;
      sa.xcd_units = from_list && (plan.P * nb) % 8 == 0 && !g_no_xcd_units;
      e = launch(kfn, from_list ? dim3(oc, plan.P, nb) : dim3(plan.P, oc, nb), dim3(kScatterThreads),
                 lds_bytes, s, sa);
      if (e != hipSuccess) return e;
      ++g_last_info[1];
      if (b0 == 0) ++g_last_info[3];
      StripCombineArgs ca;
      ca.b0 = b0; ca.oc = oc; ca.ch0 = ch0; ca.oc_total = oc_total; ca.mh = p.mh; ca.mw = p.mw;
      ca.P = plan.P; ca.slab_stride = rg.slab_stride; ca.seg_cap = sa.seg_cap; ca.fill = fill;
      ca.g_wins = l.t.wins; ca.g_unions = l.t.unions; ca.g_counts = l.t.counts; ca.g_list = l.t.list;
      ca.slabs = l.slabs; ca.out = out; ca.mask = mask;
      ca.defer_outer = sa.defer_outer; ca.head_share = sa.head_share;
      // grid.y = frames * channels <= 65535 per launch
      const int per_launch = 65535 / oc > 0 ? 65535 / oc : 1;
      if (planes) {
        PlaneCombineArgs pa;
        pa.b0 = b0; pa.oc = oc; pa.ch0 = ch0; pa.oc_total = oc_total; pa.mh = p.mh; pa.mw = p.mw;
        pa.P = plan.P; pa.plane_cap = plane_cap; pa.fill = fill;
        pa.g_counts = l.t.counts; pa.g_meta = l.t.list; pa.planes = l.slabs; pa.out = out; pa.mask = mask;
        // blocks per (frame, channel): the usual number of shared groups (about a twelfth of the windows' groups
        // at the headline geometry) at a round per thread; frames with more take further rounds
        int blocks = (plane_cap / 6 + kCombineThreads - 1) / kCombineThreads;
        if (blocks < 1) blocks = 1;
        if (blocks * kCombineThreads > plane_cap) blocks = plane_cap / kCombineThreads;      // (plane_cap: a multiple of 256)
        pa.spec_blocks = (blocks + 1) / 2;
        const dim3 g(blocks, (unsigned)(nb * oc));
        e = is_max ? launch(k_strip_combine_planes<kMax>, g, dim3(kCombineThreads), 0, s, pa)
                   : launch(k_strip_combine_planes<kMin>, g, dim3(kCombineThreads), 0, s, pa);
        if (e != hipSuccess) return e;
        ++g_last_info[2];
        continue;
      }
      for (int c0 = 0; c0 < nb; c0 += per_launch) {
        const int nc = nb - c0 < per_launch ? nb - c0 : per_launch;
        ca.b0 = b0 + c0;
        if (oc_total >= kListMinChannels) {        // (value maps of many channels: four entries per thread)
          // (at least one block per strip's list segment)
          // (many channels: the chip holds 2 K blocks at a time, so the fewer blocks per (frame, channel) the
          // fewer rounds of the same chain of round trips -- 8 blocks: 131 us at cfg3, 4: 88 us)
          const int few = kCombineSlots / kCombineEntriesMany / 2;
          const int blocks = few > plan.P ? few : plan.P;
          const dim3 g(blocks, (unsigned)(nc * oc));
          e = is_max ? launch(k_strip_combine<kMax, kCombineEntriesMany>, g, dim3(kCombineThreads), 0, s, ca)
                     : launch(k_strip_combine<kMin, kCombineEntriesMany>, g, dim3(kCombineThreads), 0, s, ca);
        } else {
          const dim3 g(kCombineSlots > plan.P ? kCombineSlots : plan.P, (unsigned)(nc * oc));
          e = is_max ? launch(k_strip_combine_one<kMax>, g, dim3(kCombineThreads), 0, s, ca)
                     : launch(k_strip_combine_one<kMin>, g, dim3(kCombineThreads), 0, s, ca);
        }
        if (e != hipSuccess) return e;
        ++g_last_info[2];
      }
    }
  }
  return hipSuccess;
}

thread_local int g_force_legacy = 0;       // dm_debug_force_legacy_window

}  // namespace

size_t strip_workspace_extra(const dm_params& p) {
  Plan plan;
  if (!make_plan(p, plan)) {       // (small batches take column strips where the cost model would cut rows: fallback_plan)
    int strips = (p.W + 39) / 40;
    if (strips > strip::kMaxStrips) strips = strip::kMaxStrips;
    if (strips < 1 || !make_plan(p, plan, strips)) return 0;
  }
  // Value maps: room for the slabs of every channel (up to 2 GiB of address space, touched only
  // where strips share cells), so that all channels go through ONE scatter + combine launch pair
  // instead of one pair per group of channels that fits the LDS-window path's 256 MiB.
  size_t more_slabs = 0;
  if (p.vc > 1) {
    const size_t per_channel = (size_t)p.B * strip::kMaxStrips / 2 * (kMaxLdsBytes / 4) * 4;
    more_slabs = per_channel * (size_t)p.vc;
    if (more_slabs > ((size_t)2 << 30)) more_slabs = (size_t)2 << 30;
  }
  return tables_bytes(p) + lists_bytes_bound(p) + 512 + more_slabs + (list_shape(p) ? pixel_list_bytes(p) : 0);
}

namespace {

// The plan for `p` with the cost model's split (strips = 0) or with a given number of strips.
const Plan* cached_plan(const dm_params& p, int strips = 0) {
  struct Slot { dm_params key; Plan plan; bool ok, valid; int strips, forced; };
  thread_local Slot slots[4] = {};
  thread_local int next = 0;
  for (Slot& s : slots)
    if (s.valid && s.strips == strips && s.forced == g_force_strips && memcmp(&s.key, &p, sizeof(dm_params)) == 0)
      return s.ok ? &s.plan : nullptr;
  Slot& s = slots[next];
  next = (next + 1) % 4;
  s.key = p; s.strips = strips; s.forced = g_force_strips; s.valid = true;
  s.ok = make_plan(p, s.plan, strips);
  return s.ok ? &s.plan : nullptr;
}

// The plan a call starts from: the cost model's split when that is column strips only; where
// the model would also cut rows or depth -- small batches, whose few workgroups the window path
// multiplies -- the strip path still takes calls of few frames (its launches carry no table
// copy): column strips of about 40 pixels.
constexpr int kSmallBatchPixels = 1 << 20;       // fallback only below this many pixels per call
const Plan* starting_plan(const dm_params& p, bool prepared) {
  const Plan* plan = cached_plan(p);
  if (plan || g_force_strips) return plan;
  if (!prepared && (int64_t)p.B * p.H * p.W > kSmallBatchPixels) return nullptr;
  int strips = (p.W + 39) / 40;
  if (strips > strip::kMaxStrips) strips = strip::kMaxStrips;
  return strips >= 1 ? cached_plan(p, strips) : nullptr;
}

// The plan and rig the call takes: the starting plan, or -- when a strip's window would not fit
// in LDS -- the same image cut into 2, 4, 8 strips.  NULL: the strip path does not apply.
const Plan* plan_and_rig(const dm_params& p, const float* pitch4, int magnitude, const Rig** rig_out,
                         bool prepared = false) {
  const Plan* plan = starting_plan(p, prepared);
  if (!plan) return nullptr;
  const Rig* rg = rig_of(p, *plan, pitch4, magnitude);
  for (int strips = 2; !rg->fits && rg->cfg.cone_ok && strips <= strip::kMaxStrips && !g_force_strips; strips *= 2) {
    if (strips <= plan->P) continue;
    const Plan* narrower = cached_plan(p, strips);
    if (!narrower) break;
    plan = narrower;
    rg = rig_of(p, *plan, pitch4, magnitude);
  }
  *rig_out = rg;
  return rg->fits ? plan : nullptr;
}

bool aligned_for_strips(const float* depth, const float* value, const uint8_t* valid, float* out, uint8_t* mask,
                        float* height, float* fused, uint8_t* fused_mask) {
  return reinterpret_cast<uintptr_t>(valid) % 4 == 0 && reinterpret_cast<uintptr_t>(out) % 16 == 0 && reinterpret_cast<uintptr_t>(mask) % 4 == 0 &&
         reinterpret_cast<uintptr_t>(fused) % 16 == 0 && reinterpret_cast<uintptr_t>(fused_mask) % 4 == 0 &&
         reinterpret_cast<uintptr_t>(height) % 16 == 0 && reinterpret_cast<uintptr_t>(depth) % 16 == 0 &&
         reinterpret_cast<uintptr_t>(value) % 16 == 0;
}

// Does the workspace hold what the call needs?  (nothing may be enqueued before the answer is yes)
bool workspace_fits(const dm_params& p, const Plan& plan, const Rig& rg, bool with_height, void* ws,
                    size_t ws_bytes, Layout& l, size_t& hm) {
  if (!carve(p, plan.P, rg.slab_stride, ws, ws_bytes, l)) return false;
  hm = with_height ? up256((size_t)p.B * p.dc * p.mh * p.mw) : 0;
  return l.slab_bytes >= hm + (size_t)p.B * plan.P * rg.slab_stride * 4;
}

// The launch sequence proper: nothing here depends on the poses but the kernel arguments (graph
// capturable: a captured launch replays the poses it was captured with, or reads them from the
// device buffer of prepared frames).
hipError_t launch_strips(const dm_params& p, const Plan& plan, const Rig& rg, const PoseSource& poses,
                         const Layout& l, size_t hm, const float* depth, const float* value,
                         const uint8_t* valid, float* out, uint8_t* mask, float* height, float* fused,
                         uint8_t* fused_mask, int* status, hipEvent_t after_projection, hipStream_t s) {
  const int oc_total = p.vc ? p.vc : p.dc;
  const bool is_max = p.reduction == DM_REDUCE_MAX;
  g_last_info[0] = g_last_info[1] = g_last_info[2] = g_last_info[3] = 0;
  hipError_t e = strip_pass(p, plan, rg, poses, l, depth, value, valid, out, mask, oc_total, p.fill, is_max,
                            l.slab_bytes - hm, fused != nullptr, status, s);
  if (e != hipSuccess) return e;
  if (height && value) {      // maps.py:332-350: second projection of the heights, NINF fill, max
    uint8_t* scratch_mask = reinterpret_cast<uint8_t*>(l.slabs) + l.slab_bytes - hm;   // (tail of the slab region)
    e = strip_pass(p, plan, rg, poses, l, depth, nullptr, valid, height, scratch_mask, p.dc, -INFINITY, true,
                   l.slab_bytes - hm, false, status, s);
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
    fa.unions = l.t.unions; fa.maps = out; fa.fused = fused; fa.fused_mask = fused_mask;
    const dim3 g((unsigned)(((size_t)p.mh * p.mw / 4 + kUnionGroups - 1) / kUnionGroups), oc_total);
    const dim3 blk(kUnionGroups * kUnionLanes);
    e = is_max ? launch(k_fuse_unions<true>, g, blk, 0, s, fa)
               : launch(k_fuse_unions<false>, g, blk, 0, s, fa);
    if (e != hipSuccess) return e;
  }
  note_split(plan.P, 1, 1, 2);
  return hipSuccess;
}

dm_frames_plan plan_of(const Plan& plan, const Rig& rg, int magnitude) {
  dm_frames_plan fp;
  memset(&fp, 0, sizeof(fp));
  fp.strips = plan.P; fp.strip_width = plan.wp; fp.slab_cells = rg.slab_stride;
  fp.max_rows = rg.max_rows; fp.max_union_cells = rg.max_union; fp.slack_cells = rg.slack;
  fp.magnitude = magnitude;
  memcpy(fp.pitch, rg.pitch, sizeof(fp.pitch));
  return fp;
}

}  // namespace

// hipErrorNotSupported: the strip path does not apply to this call (nothing enqueued).
hipError_t run_strip(const dm_params& p, const dm_frame* frames_host, const float* depth,
                     const float* value, const uint8_t* valid, float* out, uint8_t* mask,
                     float* height, float* fused, uint8_t* fused_mask, void* ws, size_t ws_bytes,
                     int* status, hipEvent_t before_projection, hipEvent_t after_projection, hipStream_t s) {
  if (g_force_legacy || p.B > 65535) return hipErrorNotSupported;
  if (!aligned_for_strips(depth, value, valid, out, mask, height, fused, fused_mask)) return hipErrorNotSupported;
  const int magnitude = validate_frames(p, frames_host, p.B);
  if (magnitude < 0) return hipErrorNotSupported;
  const float pitch4[4] = {frames_host[0].Rp[4], frames_host[0].Rp[5], frames_host[0].Rp[7], frames_host[0].Rp[8]};
  const Rig* rg = nullptr;
  const Plan* plan = plan_and_rig(p, pitch4, magnitude, &rg);
  if (!plan) return hipErrorNotSupported;
  Layout l;
  size_t hm = 0;
  if (!workspace_fits(p, *plan, *rg, height && value, ws, ws_bytes, l, hm)) return hipErrorNotSupported;
  if (before_projection) {
    const hipError_t e = hipEventRecord(before_projection, s);
    if (e != hipSuccess) return e;
  }
  const PoseSource poses = {frames_host, nullptr, p.to_global != 0};
  return launch_strips(p, *plan, *rg, poses, l, hm, depth, value, valid, out, mask, height, fused, fused_mask,
                       status, after_projection, s);
}

size_t strip_prepared_bytes(const dm_params& p) {
  return starting_plan(p, true) ? up256((size_t)p.B * sizeof(StripPose)) : 0;
}

// dm_frames_prepare_f32: validate the batch's camera state, size the launches (the plan) and --
// only when that plan is the one the caller expects, if it expects one -- upload the frames' pose
// records.  Nothing is enqueued otherwise.
hipError_t strip_prepare(const dm_params& p, const dm_frame* frames_host, void* prepared_dev,
                         size_t prepared_size, const dm_frames_plan* must_match, dm_frames_plan* plan_out,
                         hipStream_t s) {
  if (g_force_legacy || p.B > 65535 || p.B < 1) return hipErrorNotSupported;
  const int magnitude = validate_frames(p, frames_host, p.B);
  if (magnitude < 0) return hipErrorNotSupported;
  const float pitch4[4] = {frames_host[0].Rp[4], frames_host[0].Rp[5], frames_host[0].Rp[7], frames_host[0].Rp[8]};
  const Rig* rg = nullptr;
  const Plan* plan = plan_and_rig(p, pitch4, magnitude, &rg, true);
  if (!plan) return hipErrorNotSupported;
  if (reinterpret_cast<uintptr_t>(prepared_dev) % 256 != 0 || prepared_size < up256((size_t)p.B * sizeof(StripPose)))
    return hipErrorInvalidValue;
  const dm_frames_plan fp = plan_of(*plan, *rg, quantised_magnitude(magnitude));
  if (must_match && memcmp(must_match, &fp, sizeof(fp)) != 0) return hipErrorInvalidConfiguration;
  *plan_out = fp;
  thread_local std::vector<StripPose> stage;
  stage.resize(p.B);
  for (int b0 = 0; b0 < p.B; ++b0) {
    const dm_frame& f = frames_host[b0];
    StripPose& q = stage[b0];
    const bool g = p.to_global != 0;
    q.y0 = g ? f.Ry[0] : 1.0f; q.y2 = g ? f.Ry[2] : 0.0f; q.y6 = g ? f.Ry[6] : 0.0f; q.y8 = g ? f.Ry[8] : 1.0f;
    q.tx = g ? f.tx : 0.0f; q.tz = g ? f.tz : 0.0f;
    q.wo = f.width_offset; q.ho = f.height_offset; q.cam_h = f.cam_height;
    q.pad0 = q.pad1 = q.pad2 = 0.0f;
  }
  // one stream-ordered copy (the runtime has copied pageable memory out by the time it returns)
  return hipMemcpyAsync(prepared_dev, stage.data(), (size_t)p.B * sizeof(StripPose), hipMemcpyHostToDevice, s);
}

// dm_orth_project_prepared_f32
hipError_t run_strip_prepared(const dm_params& p, const dm_frames_plan& fp, const void* prepared_dev,
                              const float* depth, const float* value, const uint8_t* valid, float* out,
                              uint8_t* mask, float* height, float* fused, uint8_t* fused_mask, void* ws,
                              size_t ws_bytes, int* status, hipEvent_t before_projection,
                              hipEvent_t after_projection, hipStream_t s) {
  if (!aligned_for_strips(depth, value, valid, out, mask, height, fused, fused_mask)) return hipErrorNotSupported;
  const Plan* plan = cached_plan(p);
  if (!plan || plan->P != fp.strips) plan = cached_plan(p, fp.strips);
  if (!plan || plan->P != fp.strips || plan->wp != fp.strip_width) return hipErrorNotSupported;
  const Rig* rg = rig_of(p, *plan, fp.pitch, fp.magnitude);
  // (the plan the caller holds must be the one these parameters give: sizes of LDS tables and slabs)
  if (!rg->fits || rg->slab_stride != fp.slab_cells || rg->max_rows != fp.max_rows ||
      rg->max_union != fp.max_union_cells || rg->slack != fp.slack_cells)
    return hipErrorNotSupported;
  Layout l;
  size_t hm = 0;
  if (!workspace_fits(p, *plan, *rg, height && value, ws, ws_bytes, l, hm)) return hipErrorNotSupported;
  if (before_projection) {
    const hipError_t e = hipEventRecord(before_projection, s);
    if (e != hipSuccess) return e;
  }
  const PoseSource poses = {nullptr, static_cast<const float*>(prepared_dev), p.to_global != 0};
  return launch_strips(p, *plan, *rg, poses, l, hm, depth, value, valid, out, mask, height, fused, fused_mask,
                       status, after_projection, s);
}

// ---------------------------------------------------------------------------------------------
// dm_orth_project_fused_f32 on column strips (k_strip_fused + k_fuse_windows).
// ---------------------------------------------------------------------------------------------
namespace {

// Launch bound for strips of `wp` pixels (any number of them), every yaw and position of the
// camera: the largest window a strip can have, the largest width + height of one, the slack.
struct FusedBound {
  dm_params key;
  float pitch[4];
  int wp, mag_q;
  bool valid, ok;
  int P, slab_cells, wh_sum, h_max, slack;
  float inv, reach, g0, g1;
  float res_inv, fx_inv, fy_inv;
  bool lean;
  float fcx[8], fcz[8];         // corners of the WHOLE image's truncated frustum in the camera's local frame, cells
};

void compute_fused_bound(const dm_params& p, int wp, const float* pitch4, int mag_q, FusedBound& fb) {
  fb.key = p; memcpy(fb.pitch, pitch4, sizeof(fb.pitch)); fb.wp = wp; fb.mag_q = mag_q;
  fb.valid = true; fb.ok = false;
  if (p.reduction != DM_REDUCE_MAX && p.reduction != DM_REDUCE_MIN) return;
  if (p.mw % 4 != 0 || p.W % 4 != 0 || wp % 4 != 0 || wp < 4) return;
  if (p.mw > 32767 || p.mh > 32767 || (int64_t)p.mh * p.mw >= (1ll << 28)) return;
  if (!(p.fill == p.fill)) return;
  if (!p.has_dmin || !p.has_dmax || !(p.dmin >= 0.0f) || !(p.dmax >= p.dmin) || !isfinite(p.dmax)) return;
  if (!exact_reciprocal(p.res, &fb.res_inv) || !exact_reciprocal(p.fx, &fb.fx_inv) ||
      !exact_reciprocal(p.fy, &fb.fy_inv) || p.res < 1e-6f || p.res > 1e6f || p.fx < 1e-6f ||
      p.fx > 1e6f || p.fy < 1e-6f || p.fy > 1e6f)
    return;
  const int clip = p.clip_border > 0 ? p.clip_border : 0;
  const int r0 = clip, r1 = p.H - clip;
  if (r0 >= r1) return;
  double ay[2];
  const int rs[2] = {r0, r1 - 1};
  for (int i = 0; i < 2; ++i) {
    double yr = rs[i];
    if (p.flip_h) yr = (double)(p.H - 1) - yr;
    ay[i] = (yr - (double)p.cy) / (double)p.fy;
  }
  const float ay_lo = (float)(ay[0] < ay[1] ? ay[0] : ay[1]), ay_hi = (float)(ay[0] < ay[1] ? ay[1] : ay[0]);
  const float p5 = pitch4[1], p8 = pitch4[3];
  fb.g0 = p5 * ay_lo + p8; fb.g1 = p5 * ay_hi + p8;        // strip::cfg_rig
  if (!(fb.g0 > 1e-3f && fb.g1 > 1e-3f && strip::finite_f(fb.g0) && strip::finite_f(fb.g1))) return;
  fb.inv = (float)(1.0 / (double)p.res);
  fb.P = (p.W + wp - 1) / wp;
  fb.lean = !p.valid_c && isfinite(p.dmin) && !p.has_hmax && p.clip_border <= 0;
  const float gmax = strip::fmax2(strip::fabs_(fb.g0), strip::fabs_(fb.g1));
  float amax = 0.0f;
  std::vector<float> cx((size_t)fb.P * 8), cz((size_t)fb.P * 8);
  std::vector<char> live(fb.P);
  double rmax = 0.0;
  for (int s = 0; s < fb.P; ++s) {
    int q0 = s * wp, q1 = q0 + wp < p.W ? q0 + wp : p.W;
    if (q0 < clip) q0 = clip;
    if (q1 > p.W - clip) q1 = p.W - clip;
    live[s] = q0 < q1;
    for (int k = 0; k < 8; ++k) cx[s * 8 + k] = cz[s * 8 + k] = 0.0f;
    if (!live[s]) continue;
    const float ax_lo = (float)(((double)q0 - (double)p.cx) / (double)p.fx);
    const float ax_hi = (float)(((double)(q1 - 1) - (double)p.cx) / (double)p.fx);
    amax = strip::fmax2(amax, strip::fmax2(strip::fabs_(ax_lo), strip::fabs_(ax_hi)));
    strip::strip_corners(ax_lo, ax_hi, fb.g0, fb.g1, p.dmin, p.dmax, fb.inv, &cx[s * 8], &cz[s * 8]);
    for (int k = 0; k < 8; ++k) {
      const double r = sqrt((double)cx[s * 8 + k] * cx[s * 8 + k] + (double)cz[s * 8 + k] * cz[s * 8 + k]);
      if (r > rmax) rmax = r;
    }
  }
  fb.reach = p.dmax * fb.inv * (amax + gmax);
  {
    const int q0 = clip, q1 = p.W - clip;       // (r0 < r1 holds; a clip that eats every column leaves no live strip)
    const float ax_lo = (float)(((double)q0 - (double)p.cx) / (double)p.fx);
    const float ax_hi = (float)(((double)(q1 > q0 ? q1 - 1 : q0) - (double)p.cx) / (double)p.fx);
    strip::strip_corners(ax_lo, ax_hi, fb.g0, fb.g1, p.dmin, p.dmax, fb.inv, fb.fcx, fb.fcz);
  }
  const double slack_d = 2.0 + 16.0 * ((double)mag_q + 2.0 * (double)fb.reach) * (1.0 / 8388608.0) + 0.01;
  fb.slack = slack_d > 16.0 ? -1 : (int)ceil(slack_d);
  if (fb.slack < 0 || !isfinite(rmax)) return;
  const int steps = 1440;
  const double dtheta = 2.0 * M_PI / steps;
  const double lip = rmax * dtheta;
  const double pad_w = lip + 2.0 * fb.slack + 10.0, pad_h = lip + 2.0 * fb.slack + 4.0;
  double area = 0.0, wh = 0.0, hmax = 0.0;
  for (int i = 0; i < steps; ++i) {
    const double cs = cos(i * dtheta), sn = sin(i * dtheta);
    for (int s = 0; s < fb.P; ++s) {
      if (!live[s]) continue;
      double lx = INFINITY, hx = -INFINITY, lz = INFINITY, hz = -INFINITY;
      for (int k = 0; k < 8; ++k) {
        const double x = cs * cx[s * 8 + k] + sn * cz[s * 8 + k], z = -sn * cx[s * 8 + k] + cs * cz[s * 8 + k];
        lx = x < lx ? x : lx; hx = x > hx ? x : hx; lz = z < lz ? z : lz; hz = z > hz ? z : hz;
      }
      double w = hx - lx + pad_w, h = hz - lz + pad_h;
      if (w > p.mw) w = p.mw;
      if (h > p.mh) h = p.mh;
      if (w * h > area) area = w * h;
      if (w + h > wh) wh = w + h;
      if (h > hmax) hmax = h;
    }
  }
  fb.slab_cells = ((int)ceil(area) + 3) & ~3;
  fb.wh_sum = (int)ceil(wh);
  fb.h_max = (int)ceil(hmax);
  fb.ok = true;
}

const FusedBound* fused_bound_of(const dm_params& p, int wp, const float* pitch4, int magnitude) {
  thread_local FusedBound slots[6] = {};
  thread_local int next = 0;
  const int mag_q = quantised_magnitude(magnitude);
  for (FusedBound& f : slots)
    if (f.valid && f.wp == wp && f.mag_q == mag_q && memcmp(&f.key, &p, sizeof(dm_params)) == 0 &&
        memcmp(f.pitch, pitch4, sizeof(f.pitch)) == 0)
      return &f;
  FusedBound& f = slots[next];
  next = (next + 1) % 6;
  f = FusedBound{};
  compute_fused_bound(p, wp, pitch4, mag_q, f);
  return &f;
}

using FusedKernel = void (*)(FusedArgs);
FusedKernel pick_fused_kernel(bool is_max, bool has_valid, bool lean, bool defer) {
#define DM_F(M) {k_strip_fused<M, false, false, false>, k_strip_fused<M, true, false, false>,   \
                 k_strip_fused<M, false, true, false>, k_strip_fused<M, false, true, true>}
  static const FusedKernel table[2][4] = {DM_F(kMin), DM_F(kMax)};
#undef DM_F
  // [min | max][plain, valid map, lean, lean + deferred camera height]   (lean implies no valid map)
  return table[is_max ? 1 : 0][has_valid ? 1 : (lean ? (defer ? 3 : 2) : 0)];
}

thread_local int g_fused_force[2] = {0, 0};        // dm_debug_force_fused_split: {strip width, frames per group}
thread_local int g_last_fused[4] = {0, 0, 0, 0};   // dm_debug_last_fused_split: {strip width, strips, frames per group, groups}

}  // namespace

// hipErrorNotSupported: does not apply (nothing enqueued) -- the caller goes on to the window path.
hipError_t run_strip_fused(const dm_params& p, const dm_frame* frames_host, const float* depth,
                           const uint8_t* valid, float* out, uint8_t* mask, int accumulate, void* ws,
                           size_t ws_bytes, int* status, hipStream_t s) {
  g_last_fused[0] = g_last_fused[1] = g_last_fused[2] = g_last_fused[3] = 0;
  if (g_force_legacy || p.vc != 0 || p.B < 1 || p.B > 65535) return hipErrorNotSupported;
  if (reinterpret_cast<uintptr_t>(depth) % 16 != 0 || reinterpret_cast<uintptr_t>(valid) % 4 != 0 ||
      reinterpret_cast<uintptr_t>(out) % 16 != 0 || reinterpret_cast<uintptr_t>(mask) % 4 != 0 ||
      reinterpret_cast<uintptr_t>(ws) % 256 != 0)
    return hipErrorNotSupported;
  if ((int64_t)p.H * p.W >= (1ll << 28) || (int64_t)kFusedMaxGroup * p.dc * p.H >= (1 << 23) || p.W >= (1 << 23) ||
      (int64_t)kFusedMaxGroup * p.dc * p.H * p.W * 4 >= (1ll << 31))
    return hipErrorNotSupported;
  const int magnitude = validate_frames(p, frames_host, p.B);
  if (magnitude < 0) return hipErrorNotSupported;
  const float pitch4[4] = {frames_host[0].Rp[4], frames_host[0].Rp[5], frames_host[0].Rp[7], frames_host[0].Rp[8]};
  const bool global = p.to_global != 0;
  const double inv = 1.0 / (double)p.res;
  // the frames' camera cells and yaw directions (for the groups' pose spread and the fuse kernel's bounding box)
  thread_local std::vector<double> cam;
  cam.resize((size_t)p.B * 4);
  for (int b = 0; b < p.B; ++b) {
    const dm_frame& f = frames_host[b];
    const double xd = (global ? (double)f.tx * inv : 0.0) + (double)f.width_offset;
    double zd = (global ? (double)f.tz * inv : 0.0) + (double)f.height_offset;
    if (p.flip_h) zd = (double)(p.mh - 1) - zd;
    cam[b * 4 + 0] = xd; cam[b * 4 + 1] = zd;
    cam[b * 4 + 2] = global ? f.Ry[0] : 1.0; cam[b * 4 + 3] = global ? f.Ry[2] : 0.0;
  }
  // Candidates: groups of F = 8, 4, 2, 1 frames and the strip width that makes about one workgroup
  // per CU of it; a candidate is feasible when the windows of its groups (one frame's bound widened
  // by how far the group's poses are apart) fit in LDS and its slabs in the workspace.  Cost: the
  // scatter's waves of workgroups + the fuse kernel's slab traffic.
  struct Cand { int wp, F; const FusedBound* fb; int slab, rows; double cost; };
  Cand best = {0, 0, nullptr, 0, 0, 1e30};
  const int kWave = 256;
  for (int F = kFusedMaxGroup; F >= 1; F >>= 1) {
    if (g_fused_force[1] && F != g_fused_force[1]) continue;
    if (F > p.B && F > 1) continue;
    const int G = (p.B + F - 1) / F;
    int strips = kWave / (G * p.dc) > 0 ? kWave / (G * p.dc) : 1;
    int wp = ((p.W + strips - 1) / strips + 3) & ~3;
    if (wp < 16) wp = 16;
    if (wp > 256) wp = 256;
    if (g_fused_force[0]) wp = g_fused_force[0];
    const FusedBound* fb = fused_bound_of(p, wp, pitch4, magnitude);
    if (!fb->ok) continue;
    // how far a point can move between the first frame of a group and any other: camera cell + rotation chord
    double spread = 0.0;
    for (int g = 0; g < G && F > 1; ++g) {
      const int b0 = g * F, b1 = b0 + F < p.B ? b0 + F : p.B;
      for (int b = b0 + 1; b < b1; ++b) {
        const double m = fabs(cam[b * 4] - cam[b0 * 4]) + fabs(cam[b * 4 + 1] - cam[b0 * 4 + 1]) +
                         (double)fb->reach * hypot(cam[b * 4 + 2] - cam[b0 * 4 + 2], cam[b * 4 + 3] - cam[b0 * 4 + 3]);
        if (m > spread) spread = m;
      }
    }
    if (!(spread < 4096.0)) continue;
    const int m = F > 1 ? (int)ceil(spread) + 2 : 0;
    int64_t cells = (int64_t)fb->slab_cells + 2ll * m * fb->wh_sum + 4ll * m * m;
    if (cells > (int64_t)p.mh * p.mw) cells = (int64_t)p.mh * p.mw;
    const int slab = (int)((cells + 3) & ~3ll);
    int rows = fb->h_max + 2 * m;                       // the group window's height
    rows = ((rows < p.mh ? rows : p.mh) + 3) & ~3;
    if (fused_lds_bytes(slab, p.H, rows) > (size_t)kMaxLdsBytes) continue;
    const size_t need = up256((size_t)G * fb->P * sizeof(Win16)) + up256((size_t)G * fb->P * rows * 4) +
                        (size_t)G * p.dc * fb->P * slab * 4;
    if (need > ws_bytes) continue;
    const int wgs = fb->P * p.dc * G;
    const double waves = (double)((wgs + kWave - 1) / kWave);
    const double pixels = (double)wp * p.H * F;
    const double cost = waves * (5.0 + pixels * 2.4e-4) + 2.0 + (double)wgs * slab * 4.0 / 3.0e6;
    if (cost < best.cost) best = Cand{wp, F, fb, slab, rows, cost};
  }
  if (!best.fb) return hipErrorNotSupported;
  const FusedBound& fb = *best.fb;
  const int F = best.F, G = (p.B + F - 1) / F;
  // the fuse kernel's bounding box: every window lies within the bounding box of its frame's
  // truncated frustum (the eight corners of the whole image's, rotated by the frame's yaw as
  // strip::strip_geometry rotates a strip's) + slack
  int gx0, gx1, gz0, gz1;
  {
    double lx = INFINITY, hx = -INFINITY, lz = INFINITY, hz = -INFINITY;
    const double fs = p.flip_h ? -1.0 : 1.0;
    for (int b = 0; b < p.B; ++b) {
      const dm_frame& f = frames_host[b];
      const double y0 = global ? f.Ry[0] : 1.0, y2 = fs * (global ? f.Ry[2] : 0.0);
      const double y6 = global ? f.Ry[6] : 0.0, y8 = fs * (global ? f.Ry[8] : 1.0);
      for (int k = 0; k < 8; ++k) {
        const double x = y0 * fb.fcx[k] + y6 * fb.fcz[k] + cam[b * 4], z = y2 * fb.fcx[k] + y8 * fb.fcz[k] + cam[b * 4 + 1];
        lx = x < lx ? x : lx; hx = x > hx ? x : hx; lz = z < lz ? z : lz; hz = z > hz ? z : hz;
      }
    }
    const double R = (double)fb.slack + 3.0;
    lx = floor(lx - R); hx = ceil(hx + R) + 1.0; lz = floor(lz - R); hz = ceil(hz + R) + 1.0;
    gx0 = lx < 0.0 ? 0 : (lx > p.mw ? p.mw : (int)lx); gx1 = hx > p.mw ? p.mw : (hx < 0.0 ? 0 : (int)hx);
    gz0 = lz < 0.0 ? 0 : (lz > p.mh ? p.mh : (int)lz); gz1 = hz > p.mh ? p.mh : (hz < 0.0 ? 0 : (int)hz);
    gx0 &= ~3; gx1 = (gx1 + 3) & ~3;
    if (gx1 > p.mw) gx1 = p.mw;
    if (gx1 <= gx0 || gz1 <= gz0) { gx0 = gx1 = gz0 = gz1 = 0; }
  }
  const bool is_max = p.reduction == DM_REDUCE_MAX;
  // (the camera height deferred to the flush: only without a height truncation, which compares y1)
  bool defer = fb.lean;
  for (int b = 1; b < p.B; ++b) defer = defer && frames_host[b].cam_height == frames_host[0].cam_height;
  const FusedKernel kfn = pick_fused_kernel(is_max, valid != nullptr, fb.lean, defer);
  hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn));
  if (e != hipSuccess) return e;
  Win16* wins = static_cast<Win16*>(ws);
  uint32_t* spans = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(ws) + up256((size_t)G * fb.P * sizeof(Win16)));
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(spans) + up256((size_t)G * fb.P * best.rows * 4));
  thread_local FusedArgs fa;
  memset(&fa, 0, offsetof(FusedArgs, poses));
  fa.W = p.W; fa.H = p.H; fa.clip = p.clip_border > 0 ? p.clip_border : 0; fa.flip_h = p.flip_h != 0;
  fa.cx = p.cx; fa.cy = p.cy; fa.fx = p.fx; fa.fy = p.fy; fa.res = p.res;
  fa.fx_inv = fb.fx_inv; fa.fy_inv = fb.fy_inv; fa.res_inv = fb.res_inv;
  fa.dmin = p.dmin; fa.dmax = p.dmax; fa.hmax = p.has_hmax ? p.hmax : INFINITY;
  fa.Hm1 = (float)(p.H - 1); fa.mhm1 = (float)(p.mh - 1);
  fa.p4 = pitch4[0]; fa.p5 = pitch4[1]; fa.p7 = pitch4[2]; fa.p8 = pitch4[3];
  fa.wp = best.wp; fa.P = fb.P; fa.F = F;
  fa.dc = p.dc; fa.valid_c = p.valid_c; fa.slab_stride = best.slab; fa.max_rows = best.rows; fa.mh = p.mh; fa.mw = p.mw;
  fa.fill = p.fill; fa.cam_h = frames_host[0].cam_height;
  fa.inv = fb.inv; fa.reach = fb.reach; fa.g0 = fb.g0; fa.g1 = fb.g1; fa.cone_ok = 1;
  fa.slabs = slabs; fa.wins = wins; fa.spans = spans; fa.status = status;
#ifdef DM_STAMPS
  fa.stamps = g_stamp_buffer;
#endif
  const size_t lds_bytes = fused_lds_bytes(best.slab, p.H, best.rows);
  const size_t N = (size_t)p.H * p.W;
  const int per_launch = kPoseFrames / F * F;           // whole groups per launch
  for (int b0 = 0; b0 < p.B; b0 += per_launch) {
    const int nb = p.B - b0 < per_launch ? p.B - b0 : per_launch;
    fa.nb = nb; fa.group0 = b0 / F;
    fa.depth = depth + (size_t)b0 * p.dc * N;
    fa.valid = valid ? valid + (size_t)b0 * p.valid_c * N : nullptr;
    for (int i = 0; i < nb; ++i) {
      const dm_frame& f = frames_host[b0 + i];
      StripPose& q = fa.poses[i];
      q.y0 = global ? f.Ry[0] : 1.0f; q.y2 = global ? f.Ry[2] : 0.0f; q.y6 = global ? f.Ry[6] : 0.0f; q.y8 = global ? f.Ry[8] : 1.0f;
      q.tx = global ? f.tx : 0.0f; q.tz = global ? f.tz : 0.0f;
      q.wo = f.width_offset; q.ho = f.height_offset; q.cam_h = f.cam_height;
      q.pad0 = q.pad1 = q.pad2 = 0.0f;
    }
    e = launch(kfn, dim3(fb.P, p.dc, (nb + F - 1) / F), dim3(kScatterThreads), lds_bytes, s, fa);
    if (e != hipSuccess) return e;
  }
  FuseWinArgs fw;
  fw.nwin = G * fb.P; fw.b0 = 0; fw.nparts = fb.P; fw.oc = p.dc; fw.ch0 = 0; fw.oc_total = p.dc;
  fw.mh = p.mh; fw.mw = p.mw; fw.slab_stride = best.slab; fw.accumulate = accumulate; fw.fill = p.fill;
  fw.gx0 = gx0; fw.gz0 = gz0; fw.gx1 = gx1; fw.gz1 = gz1;
  fw.wins = wins; fw.slabs = slabs; fw.fused = out; fw.fused_mask = mask;
  fw.spans = spans; fw.span_rows = best.rows;
  const int bw4 = (gx1 - gx0) / 4;
  const int heavy = gx1 > gx0 ? ((bw4 + kFuseGroups - 1) / kFuseGroups) * (gz1 - gz0) : 0;
  const int per_fill_block = kFuseGroups * kFuseLanes * 8;
  const int fill_blocks = (int)(((size_t)p.mh * p.mw / 4 + per_fill_block - 1) / per_fill_block);
  const dim3 g((unsigned)(heavy + fill_blocks), p.dc);
  const dim3 blk(kFuseGroups * kFuseLanes);
  e = is_max ? launch(k_fuse_windows<true>, g, blk, 0, s, fw) : launch(k_fuse_windows<false>, g, blk, 0, s, fw);
  if (e != hipSuccess) return e;
  g_last_fused[0] = best.wp; g_last_fused[1] = fb.P; g_last_fused[2] = F; g_last_fused[3] = G;
  note_split(fb.P, 1, 1, 2);
  return hipSuccess;
}

}  // namespace dm

extern "C" __attribute__((visibility("default"))) void dm_debug_force_fused_split(int strip_width, int frames_per_group) {
  dm::g_fused_force[0] = strip_width > 0 ? (strip_width + 3) & ~3 : 0;
  dm::g_fused_force[1] = frames_per_group > 0 ? frames_per_group : 0;
}

extern "C" __attribute__((visibility("default"))) void dm_debug_last_fused_split(int32_t* out4) {
  for (int i = 0; i < 4; ++i) out4[i] = dm::g_last_fused[i];
}

extern "C" __attribute__((visibility("default"))) int dm_debug_force_strips(int strips) {
  const int old = dm::g_force_strips;
  dm::g_force_strips = strips > 0 && strips <= dm::strip::kMaxStrips ? strips : 0;
  return old;
}

extern "C" __attribute__((visibility("default"))) int dm_debug_fill_split(int head_share) {
  const int old = dm::g_fill_split;
  dm::g_fill_split = head_share < 0 ? -1 : (head_share > 8 ? 8 : head_share);
  return old;
}

extern "C" __attribute__((visibility("default"))) int dm_debug_force_nt_fill(int mode) {
  const int old = dm::g_force_nt_fill;
  dm::g_force_nt_fill = mode < 0 ? -1 : (mode != 0);
  return old;
}

extern "C" __attribute__((visibility("default"))) int dm_debug_strip_value_list(int on) {
  const int old = !dm::g_no_value_list;
  dm::g_no_value_list = on == 0;
  return old;
}

extern "C" __attribute__((visibility("default"))) size_t dm_debug_strip_slab_budget(size_t bytes) {
  const size_t old = dm::g_strip_slab_budget;
  dm::g_strip_slab_budget = bytes;
  return old;
}

extern "C" __attribute__((visibility("default"))) void dm_debug_last_strip_info(int32_t* out4) {
  for (int i = 0; i < 4; ++i) out4[i] = dm::g_last_info[i];
}

extern "C" __attribute__((visibility("default"))) int dm_debug_planes(int mode) {
  const int old = dm::g_planes;
  dm::g_planes = mode < 0 ? -1 : (mode != 0);
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
  const int magnitude = validate_frames(*p, frames, p->B);
  if (out_bound) out_bound[0] = out_bound[1] = out_bound[2] = out_bound[3] = out_bound[4] = 0;
  if (magnitude < 0) {
    if (out_bound) out_bound[0] = -1;
    for (int b = 0; b < p->B; ++b) memset(out_geom + (size_t)b * (8 + 4 * strip::kMaxStrips), 0, (8 + 4 * strip::kMaxStrips) * 4);
    return plan.P;
  }
  Rig rg = {};
  const float pitch4[4] = {frames[0].Rp[4], frames[0].Rp[5], frames[0].Rp[7], frames[0].Rp[8]};
  compute_rig(*p, plan, pitch4, quantised_magnitude(magnitude), rg);
  if (out_bound) {
    out_bound[0] = rg.slack; out_bound[1] = rg.fits; out_bound[2] = rg.slab_stride;
    out_bound[3] = rg.max_rows; out_bound[4] = rg.max_union;
  }
  const int stride = 8 + 4 * strip::kMaxStrips;
  for (int b = 0; b < p->B; ++b) {
    dm_frame f = frames[b];
    if (!p->to_global) {      // as passed to the device: neutral yaw, no translation
      static const float eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      memcpy(f.Ry, eye, sizeof(eye)); f.tx = 0.0f; f.tz = 0.0f;
    }
    strip::FrameGeom g;
    strip::frame_geometry(rg.cfg, f.Rp, g);
    int32_t* o = out_geom + (size_t)b * stride;
    o[0] = g.ok | (g.inside << 8); o[1] = plan.P; o[2] = plan.wp; o[3] = rg.slack;
    o[4] = g.U.x0; o[5] = g.U.z0; o[6] = g.U.w; o[7] = g.U.h;
    for (int s = 0; s < strip::kMaxStrips; ++s) {
      o[8 + 4 * s] = g.win[s].x0; o[9 + 4 * s] = g.win[s].z0;
      o[10 + 4 * s] = g.win[s].w; o[11 + 4 * s] = g.win[s].h;
    }
    if (out_covers) {
      // (B, mh, P, 2): cover and owned span of every strip on every map row
      uint32_t cover[strip::kMaxStrips];
      const int P2 = plan.P <= 4 ? 4 : 8;
      for (int z = 0; z < p->mh; ++z) {
        for (int s = 0; s < plan.P; ++s) cover[s] = strip::row_cover(g.win[s], g.L[s], g.R[s], z, p->mw);
        for (int s = plan.P; s < strip::kMaxStrips; ++s) cover[s] = 0u;
        for (int s = 0; s < plan.P; ++s) {
          uint32_t* o2 = out_covers + (((size_t)b * p->mh + z) * plan.P + s) * 2;
          o2[0] = cover[s];
          o2[1] = strip::row_owned(cover, s, plan.P, P2);
        }
      }
    }
  }
  return plan.P;
}

// GPU: the same geometry from k_strip_geometry_dump (the device's lane-parallel evaluation, the
// function the scatter kernel's head runs), for the test that host and device agree bit for bit.
// frames_dev (B, 32) f32 (a local map's records with a neutral yaw), frames_host: the same records
// on the host (for the rig); geom_dev: B * sizeof(FrameGeom) + B * 48 bytes of device scratch.
extern "C" __attribute__((visibility("default"))) int dm_debug_strip_geometry_dev(
    const dm_params* p, const dm_frame* frames_host, const float* frames_dev, void* geom_dev,
    size_t geom_bytes, void* stream) {
  using namespace dm;
  if (!p || !frames_host || !frames_dev || !geom_dev || p->B < 1) return -1;
  Plan plan;
  if (!make_plan(*p, plan)) return 0;
  const int magnitude = validate_frames(*p, frames_host, p->B);
  if (magnitude < 0) return 0;
  Rig rg = {};
  const float pitch4[4] = {frames_host[0].Rp[4], frames_host[0].Rp[5], frames_host[0].Rp[7], frames_host[0].Rp[8]};
  compute_rig(*p, plan, pitch4, quantised_magnitude(magnitude), rg);
  const size_t need = (size_t)p->B * sizeof(strip::FrameGeom) + (size_t)p->B * sizeof(StripPose);
  if (geom_bytes < need) return -(int)need;
  float* poses_dev = reinterpret_cast<float*>(static_cast<unsigned char*>(geom_dev) + (size_t)p->B * sizeof(strip::FrameGeom));
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(k_pose_records, dim3((p->B + 63) / 64), dim3(64), 0, s, frames_dev, poses_dev, p->B);
  hipLaunchKernelGGL(k_strip_geometry_dump, dim3(p->B), dim3(64), 0, s, rg.args, poses_dev,
                     static_cast<strip::FrameGeom*>(geom_dev));
  return hipGetLastError() == hipSuccess ? plan.P : -2;
}

#ifdef DM_STAMPS
extern "C" __attribute__((visibility("default"))) void dm_debug_strip_stamp_buffer(long long* dev) {
  dm::g_stamp_buffer = dev;
}
#endif
