// Strip path of orth_project (max / min): host side.  See dm_strip_geometry.hpp for the
// geometry and dm_strip_kernels.hpp for the kernels.
//
// What the host does per call is pose independent: it validates the frame records (the
// rotations have the axis-aligned pattern the fast arithmetic needs, one camera pitch for the
// whole batch, magnitudes within the float32 slack), copies them to the device (or takes
// them from the device as they are: dm_frames_prepare_f32 / dm_orth_project_prepared_f32)
// and launches k_strip_prepare (the frames' geometry and row tables, on the device: once per
// set of poses for prepared frames) and k_strip_scatter + k_strip_merge with sizes taken from
// a bound that holds for EVERY yaw and position of the camera (cached per camera rig).
#include <math.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "dm_kernels.hpp"
#include "dm_strip_kernels.hpp"

// Workgroups per launch the strip path aims at by adding fill-only workgroups (0: none).  Measured
// at 16 x 1280x960 -> 2048^2 (8 strips per frame): 8 + 8 workgroups per frame 120 us, 8 + 24: 142 us,
// against 97-108 us on the window path with its row blocks -- so off, and such shapes stay there.
#ifndef DM_X_COMBINE_ENTRIES
#define DM_X_COMBINE_ENTRIES 4      // list entries per thread of the combine kernel for value maps of many channels
#endif
#ifndef DM_X_FILL_TARGET
#define DM_X_FILL_TARGET 0
#endif

namespace dm {

void note_split(int pc, int pr, int pd, int path);     // dm_window.hip (dm_debug_last_split)

namespace {

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }
constexpr size_t kCfgBytes = 1024;
static_assert(sizeof(strip::Cfg) <= kCfgBytes, "Cfg slot");

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
  bool valid, fits;
  strip::Cfg cfg;
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

// The frame records the strip path can take: axis-aligned rotations (the fast arithmetic),
// one pitch for the batch, finite values; returns the cells of slack that cover every frame
// (>= what the device computes for it) or -1.
int validate_frames(const dm_params& p, const dm_frame* f, int B) {
  double worst = 0.0;
  const double inv = 1.0 / (double)p.res;
  for (int b = 0; b < B; ++b) {
    const float* rp = f[b].Rp;
    const float* ry = f[b].Ry;
    if (!(rp[0] == 1.0f && rp[1] == 0.0f && rp[2] == 0.0f && rp[3] == 0.0f && rp[6] == 0.0f)) return -1;
    if (p.to_global &&
        !(ry[1] == 0.0f && ry[3] == 0.0f && ry[4] == 1.0f && ry[5] == 0.0f && ry[7] == 0.0f))
      return -1;
    if (rp[4] != f[0].Rp[4] || rp[5] != f[0].Rp[5] || rp[7] != f[0].Rp[7] || rp[8] != f[0].Rp[8]) return -1;
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
void compute_rig(const dm_params& p, const Plan& plan, const dm_frame& f0, int magnitude, Rig& rg) {
  rg.key = p; rg.P = plan.P; rg.valid = true; rg.fits = false;
  rg.pitch[0] = f0.Rp[4]; rg.pitch[1] = f0.Rp[5]; rg.pitch[2] = f0.Rp[7]; rg.pitch[3] = f0.Rp[8];
  strip::Cfg& c = rg.cfg;
  memset(&c, 0, sizeof(c));
  c.P = plan.P; c.mw = p.mw; c.mh = p.mh; c.flip_h = p.flip_h != 0;
  c.inv = (float)(1.0 / (double)p.res);
  memcpy(c.live, plan.live, sizeof(c.live));
  strip::cfg_rig(c, f0.Rp, plan.ax_lo, plan.ax_hi, plan.ay_lo, plan.ay_hi, p.dmin, p.dmax);
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
  rg.max_rows = (int)ceil(uh);
  rg.max_union = (((int)ceil(uw) + 2 * strip::kSpanAlign + 3) & ~3) * rg.max_rows;
  rg.fits = strip_lds_bytes(rg.slab_stride, rg.max_rows, p.H) <= (size_t)kMaxLdsBytes;
}

const Rig* rig_of(const dm_params& p, const Plan& plan, const dm_frame& f0, int magnitude) {
  thread_local Rig slots[4] = {};
  thread_local int slot_mag[4] = {0, 0, 0, 0};
  thread_local int next = 0;
  // (the magnitude only matters through the slack it implies: reuse a rig computed for a
  // magnitude at least as large and at most 2^20 larger -- the same slack up to 2 cells)
  for (int i = 0; i < 4; ++i) {
    Rig& r = slots[i];
    if (r.valid && r.P == plan.P && memcmp(&r.key, &p, sizeof(dm_params)) == 0 &&
        r.pitch[0] == f0.Rp[4] && r.pitch[1] == f0.Rp[5] && r.pitch[2] == f0.Rp[7] && r.pitch[3] == f0.Rp[8] &&
        magnitude <= slot_mag[i] && slot_mag[i] - magnitude < (1 << 19))
      return &r;
  }
  Rig& r = slots[next];
  const int mag_up = magnitude < (1 << 18) ? (1 << 18) : magnitude + (1 << 17);   // a little headroom
  slot_mag[next] = mag_up;
  next = (next + 1) % 4;
  r = Rig{};
  compute_rig(p, plan, f0, mag_up, r);
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


// Value maps of kListMinChannels channels or more (one depth channel for all of them): the cell of
// every pixel is computed once (index pass) and the channels scatter from that list (value pass).
constexpr int kListMinChannels = 3;
StripKernel pick_index_kernel(bool has_valid, bool lean) {
  if (lean && !has_valid) return k_strip_scatter<kMax, false, false, true, kIndexOut>;
  return has_valid ? k_strip_scatter<kMax, true, false, false, kIndexOut>
                   : k_strip_scatter<kMax, false, false, false, kIndexOut>;
}
StripKernel pick_value_kernel(bool is_max) {
  return is_max ? k_strip_scatter<kMax, false, true, true, kFromList>
                : k_strip_scatter<kMin, false, true, true, kFromList>;
}
inline size_t pixel_list_bytes(const dm_params& p) {     // (B, P, H, wp) uint16: P * wp <= W + 32 per strip
  return up256((size_t)p.B * p.H * ((size_t)p.W + 32 * strip::kMaxStrips) * 2);
}
thread_local int g_no_value_list = 0;      // dm_debug_strip_value_list(0): every channel recomputes its cells
inline bool list_shape(const dm_params& p) { return p.vc >= kListMinChannels && p.dc == 1; }   // (sizes the workspace)
inline bool wants_pixel_list(const dm_params& p) { return list_shape(p) && !g_no_value_list; }

// Device copy of a batch's camera state as the kernels read it ("prepared frames"):
// [Cfg (kCfgBytes) | status word (256 B) | list counters | frame records | frame tables].
// Lives at the head of the workspace for dm_orth_project_f32 (staged by one copy per call, the
// tables by k_strip_prepare right behind it) or in a buffer of the caller's that
// dm_frames_prepare_f32 filled once (dm_orth_project_prepared_f32: no copy and no geometry,
// host or device).
struct PreparedView {
  const strip::Cfg* cfg;
  int* status;
  const float* frames;        // (B, 32)
  FrameTables t;              // sized from the parameters alone; indexed with the plan's max_rows, P, list_cap
};
constexpr size_t kStatusBytes = 256;
// entries of a frame's shared-group list: every float4 group of the largest union window
inline size_t list_cap_bound(const dm_params& p) {
  const size_t rows = p.mh < kListMaxRows ? p.mh : kListMaxRows;
  const size_t groups = p.mw / 4 < kListMaxGroups ? p.mw / 4 : kListMaxGroups;
  return rows * groups;
}
inline size_t tables_bytes(const dm_params& p) {
  return up256((size_t)p.B * strip::kMaxStrips * sizeof(Win16)) + up256((size_t)p.B * sizeof(Win16)) +
         up256((size_t)p.B * sizeof(int)) +
         up256((size_t)p.B * p.mh * strip::kMaxStrips * sizeof(strip::RowEntry)) +
         up256((size_t)p.B * p.mh * sizeof(uint2)) + up256((size_t)p.B * list_cap_bound(p) * sizeof(uint32_t));
}
inline size_t counts_bytes(const dm_params& p) { return up256((size_t)p.B * sizeof(int)); }
inline size_t staged_bytes(const dm_params& p) {      // what the host copies: Cfg, status, list counters (zero), frame records
  return kCfgBytes + kStatusBytes + counts_bytes(p) + up256((size_t)p.B * sizeof(dm_frame));
}
inline size_t prepared_bytes(const dm_params& p) { return staged_bytes(p) + tables_bytes(p); }
inline PreparedView view_prepared(const dm_params& p, void* dev) {
  unsigned char* base = static_cast<unsigned char*>(dev);
  PreparedView v;
  v.cfg = reinterpret_cast<const strip::Cfg*>(base);
  v.status = reinterpret_cast<int*>(base + kCfgBytes);
  v.t.counts = reinterpret_cast<int*>(base + kCfgBytes + kStatusBytes);
  v.frames = reinterpret_cast<const float*>(base + kCfgBytes + kStatusBytes + counts_bytes(p));
  base += staged_bytes(p);
  v.t.wins = reinterpret_cast<Win16*>(base); base += up256((size_t)p.B * strip::kMaxStrips * sizeof(Win16));
  v.t.unions = reinterpret_cast<Win16*>(base); base += up256((size_t)p.B * sizeof(Win16));
  v.t.flags = reinterpret_cast<int*>(base); base += up256((size_t)p.B * sizeof(int));
  v.t.rows = reinterpret_cast<strip::RowEntry*>(base); base += up256((size_t)p.B * p.mh * strip::kMaxStrips * sizeof(strip::RowEntry));
  v.t.reach = reinterpret_cast<uint2*>(base); base += up256((size_t)p.B * p.mh * sizeof(uint2));
  v.t.list = reinterpret_cast<uint32_t*>(base);
  return v;
}
// entries per frame of the shared-group lists of a plan
inline int list_cap_of(const dm_params& p, const dm_frames_plan& fp) {
  const size_t cap = (size_t)fp.max_union_cells / 4 + 1;
  return (int)(cap < list_cap_bound(p) ? cap : list_cap_bound(p));
}

struct Layout {               // workspace of the strip path
  float* slabs;
  size_t slab_bytes;
  uint16_t* pixel_list;       // value maps of many channels: the pixels' cells (or NULL)
};

// [256 B | slabs ... | pixel list]
bool carve(const dm_params& p, void* ws, size_t ws_bytes, Layout& l) {
  if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return false;
  ws_bytes = ws_bytes / 256 * 256;
  const size_t lb = list_shape(p) ? pixel_list_bytes(p) : 0;
  if (ws_bytes < 512 + lb) return false;
  l.slabs = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + 256);
  l.slab_bytes = ws_bytes - 256 - lb;
  l.pixel_list = lb ? reinterpret_cast<uint16_t*>(static_cast<unsigned char*>(ws) + ws_bytes - lb) : nullptr;
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
hipError_t strip_pass(const dm_params& p, const Plan& plan, const dm_frames_plan& rb, const PreparedView& pv,
                      const Layout& l,
                      const float* depth, const float* value, const uint8_t* valid, float* out,
                      uint8_t* mask, int oc_total, float fill, bool is_max, size_t slab_bytes,
                      hipStream_t s) {
  StripArgs sa;
  memset(&sa, 0, sizeof(sa));
  sa.W = p.W; sa.H = p.H;
  sa.clip = p.clip_border > 0 ? p.clip_border : 0;
  sa.flip_h = p.flip_h != 0;
  sa.cx = p.cx; sa.cy = p.cy; sa.fx = p.fx; sa.fy = p.fy; sa.res = p.res;
  sa.res_inv = plan.res_inv; sa.fx_inv = plan.fx_inv; sa.fy_inv = plan.fy_inv;
  sa.dmin = p.dmin; sa.dmax = p.dmax;
  sa.hmax = p.has_hmax ? p.hmax : INFINITY;
  sa.Hm1 = (float)(p.H - 1); sa.mhm1 = (float)(p.mh - 1);
  sa.wp = plan.wp; sa.P = plan.P;
  // (experiment, off: fill-only workgroups beside the strips' for small batches, DM_X_FILL_TARGET)
  {
    const long units = (long)p.B * oc_total;
    long f = (DM_X_FILL_TARGET + units - 1) / units;
    if (f > 4 * plan.P) f = 4 * plan.P;
    if (f > p.mh / 16) f = p.mh / 16;
    sa.fill_parts = f > plan.P ? (int)f : plan.P;
  }
  sa.dc = p.dc; sa.valid_c = p.valid_c;
  sa.oc_total = oc_total;
  sa.slab_stride = rb.slab_cells;
  sa.max_rows = rb.max_rows;
  sa.fill = fill;
  sa.b0 = 0;
  sa.frames = pv.frames;
  sa.depth = depth; sa.value = value; sa.valid = valid;
  sa.slabs = l.slabs;
  sa.out = out; sa.mask = mask; sa.mh = p.mh; sa.mw = p.mw;
  sa.g_wins = pv.t.wins; sa.g_unions = pv.t.unions; sa.g_flags = pv.t.flags; sa.g_rows = pv.t.rows;
  sa.g_reach = pv.t.reach;
#ifdef DM_STAMPS
  sa.stamps = g_stamp_buffer;
#endif
  const bool has_valid = valid != nullptr, has_value = value != nullptr;
  const bool from_list = has_value && l.pixel_list != nullptr && wants_pixel_list(p);
  sa.list = from_list ? l.pixel_list : nullptr;
  const StripKernel kfn = from_list ? pick_value_kernel(is_max)
                                    : pick_strip_kernel(is_max, has_valid, has_value, plan.lean);
  hipError_t e = raise_lds_limit(reinterpret_cast<const void*>(kfn));
  if (e != hipSuccess) return e;
  const size_t lds_bytes = strip_lds_bytes(rb.slab_cells, rb.max_rows, p.H);
  if (lds_bytes > (size_t)kMaxLdsBytes) return hipErrorNotSupported;
  if (from_list) {            // the index pass: every pixel's cell inside its strip's window, once for all channels
    const StripKernel ifn = pick_index_kernel(has_valid, plan.lean);
    e = raise_lds_limit(reinterpret_cast<const void*>(ifn));
    if (e != hipSuccess) return e;
    StripArgs ia = sa;
    ia.value = nullptr; ia.out = nullptr; ia.mask = nullptr; ia.oc = 1; ia.ch0 = 0; ia.oc_total = 1;
    ia.fill_parts = plan.P;
    e = launch(ifn, dim3(plan.P, 1, p.B), dim3(kScatterThreads), lds_bytes, s, ia);
    if (e != hipSuccess) return e;
  }
  // channel groups: the slabs of one group fit the slab region
  const size_t per_channel = (size_t)p.B * plan.P * rb.slab_cells * 4;
  int group = (int)(slab_bytes / (per_channel ? per_channel : 1));
  if (group < 1) return hipErrorNotSupported;
  if (group > oc_total) group = oc_total;
  if (group > 65535) group = 65535;
  for (int ch0 = 0; ch0 < oc_total; ch0 += group) {
    const int oc = oc_total - ch0 < group ? oc_total - ch0 : group;
    sa.oc = oc; sa.ch0 = ch0;
    e = launch(kfn, from_list ? dim3(oc, sa.fill_parts, p.B) : dim3(sa.fill_parts, oc, p.B), dim3(kScatterThreads),
               lds_bytes, s, sa);
    if (e != hipSuccess) return e;
    StripCombineArgs ca;
    ca.b0 = 0; ca.oc = oc; ca.ch0 = ch0; ca.oc_total = oc_total; ca.mh = p.mh; ca.mw = p.mw;
    ca.P = plan.P; ca.slab_stride = rb.slab_cells; ca.list_cap = list_cap_of(p, rb); ca.fill = fill;
    ca.g_wins = pv.t.wins; ca.g_unions = pv.t.unions; ca.g_counts = pv.t.counts; ca.g_list = pv.t.list;
    ca.slabs = l.slabs; ca.out = out; ca.mask = mask;
    // grid.y = frames * channels <= 65535 per launch
    const int per_launch = 65535 / oc > 0 ? 65535 / oc : 1;
    for (int b0 = 0; b0 < p.B; b0 += per_launch) {
      const int nb = p.B - b0 < per_launch ? p.B - b0 : per_launch;
      ca.b0 = b0;
      if (oc_total >= kListMinChannels) {        // (value maps of many channels: four entries per thread)
        const dim3 g(kCombineSlots / DM_X_COMBINE_ENTRIES, (unsigned)(nb * oc));
        e = is_max ? launch(k_strip_combine<kMax, DM_X_COMBINE_ENTRIES>, g, dim3(kCombineThreads), 0, s, ca)
                   : launch(k_strip_combine<kMin, DM_X_COMBINE_ENTRIES>, g, dim3(kCombineThreads), 0, s, ca);
      } else {
        const dim3 g(kCombineSlots, (unsigned)(nb * oc));
        e = is_max ? launch(k_strip_combine_one<kMax>, g, dim3(kCombineThreads), 0, s, ca)
                   : launch(k_strip_combine_one<kMin>, g, dim3(kCombineThreads), 0, s, ca);
      }
      if (e != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

thread_local int g_force_legacy = 0;       // dm_debug_force_legacy_window

}  // namespace

size_t strip_workspace_extra(const dm_params& p) {
  Plan plan;
  if (!make_plan(p, plan)) {       // (prepared frames take column strips where a plain call would not: prepare_plan)
    int strips = (p.W + 39) / 40;
    if (strips > strip::kMaxStrips) strips = strip::kMaxStrips;
    if (strips < 1 || !make_plan(p, plan, strips)) return 0;
  }
  // Value maps: room for the slabs of every channel (up to 2 GiB of address space, touched only
  // where strips share cells), so that all channels go through ONE scatter + merge launch pair
  // instead of one pair per group of channels that fits the LDS-window path's 256 MiB.
  size_t more_slabs = 0;
  if (p.vc > 1) {
    const size_t per_channel = (size_t)p.B * strip::kMaxStrips / 2 * (kMaxLdsBytes / 4) * 4;
    more_slabs = per_channel * (size_t)p.vc;
    if (more_slabs > ((size_t)2 << 30)) more_slabs = (size_t)2 << 30;
  }
  return prepared_bytes(p) + 512 + more_slabs + (list_shape(p) ? pixel_list_bytes(p) : 0);
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

// The plan and rig the call takes: the cost model's split, or -- when a strip's window would not
// fit in LDS -- the same image cut into 2, 4, 8 strips.  NULL: the strip path does not apply.
// A request to PREPARE frames takes column strips even where the cost model of a plain call would
// split rows or depth (a single small frame: the plain call is cheaper on the window path, whose
// geometry comes from the host; prepared, the strip path's two launches are all there is): about
// 40 columns per strip.
const Plan* prepare_plan(const dm_params& p) {
  const Plan* plan = cached_plan(p);
  if (plan || g_force_strips) return plan;
  int strips = (p.W + 39) / 40;
  if (strips > strip::kMaxStrips) strips = strip::kMaxStrips;
  return strips >= 1 ? cached_plan(p, strips) : nullptr;
}

const Plan* plan_and_rig(const dm_params& p, const dm_frame* frames_host, const Rig** rig_out, bool for_prepare = false) {
  const int magnitude = validate_frames(p, frames_host, p.B);
  if (magnitude < 0) return nullptr;
  const Plan* plan = for_prepare ? prepare_plan(p) : cached_plan(p);
  if (!plan) return nullptr;
  const Rig* rg = rig_of(p, *plan, frames_host[0], magnitude);
  for (int strips = 2; !rg->fits && rg->cfg.cone_ok && strips <= strip::kMaxStrips && !g_force_strips; strips *= 2) {
    if (strips <= plan->P) continue;
    const Plan* narrower = cached_plan(p, strips);
    if (!narrower) break;
    plan = narrower;
    rg = rig_of(p, *plan, frames_host[0], magnitude);
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

// [Cfg | status | frame records] of a batch as one block of host memory (thread-local staging).
const std::vector<unsigned char>& stage_prepared(const dm_params& p, const Rig& rg, const dm_frame* frames_host) {
  thread_local std::vector<unsigned char> stage;
  stage.resize(staged_bytes(p));
  memset(stage.data(), 0, kCfgBytes + kStatusBytes + counts_bytes(p));
  memcpy(stage.data(), &rg.cfg, sizeof(strip::Cfg));
  dm_frame* f = reinterpret_cast<dm_frame*>(stage.data() + kCfgBytes + kStatusBytes + counts_bytes(p));
  memcpy(f, frames_host, (size_t)p.B * sizeof(dm_frame));
  if (!p.to_global) {       // local map: neutral yaw, no translation (exact: x * 1 + z * 0 + 0)
    static const float eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int b = 0; b < p.B; ++b) { memcpy(f[b].Ry, eye, sizeof(eye)); f[b].tx = 0.0f; f[b].tz = 0.0f; }
  }
  return stage;
}

// The launch sequence proper: nothing here depends on the poses (graph capturable).
hipError_t launch_strips(const dm_params& p, const Plan& plan, const dm_frames_plan& fp, const PreparedView& pv,
                         const float* depth, const float* value, const uint8_t* valid, float* out,
                         uint8_t* mask, float* height, float* fused, uint8_t* fused_mask, void* ws,
                         size_t ws_bytes, hipEvent_t after_projection, hipStream_t s) {
  const int oc_total = p.vc ? p.vc : p.dc;
  Layout l;
  if (!carve(p, ws, ws_bytes, l)) return hipErrorNotSupported;
  const size_t hm = (height && value) ? up256((size_t)p.B * p.dc * p.mh * p.mw) : 0;
  if (l.slab_bytes < hm + (size_t)p.B * plan.P * fp.slab_cells * 4) return hipErrorNotSupported;
  const bool is_max = p.reduction == DM_REDUCE_MAX;
  hipError_t e = strip_pass(p, plan, fp, pv, l, depth, value, valid, out, mask, oc_total, p.fill, is_max,
                            l.slab_bytes - hm, s);
  if (e != hipSuccess) return e;
  if (height && value) {      // maps.py:332-350: second projection of the heights, NINF fill, max
    uint8_t* scratch_mask = reinterpret_cast<uint8_t*>(l.slabs) + l.slab_bytes - hm;   // (tail of the slab region)
    e = strip_pass(p, plan, fp, pv, l, depth, nullptr, valid, height, scratch_mask, p.dc, -INFINITY, true,
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
    fa.unions = pv.t.unions; fa.maps = out; fa.fused = fused; fa.fused_mask = fused_mask;
    const dim3 g((unsigned)(((size_t)p.mh * p.mw / 4 + kFuseGroups - 1) / kFuseGroups), oc_total);
    const dim3 blk(kFuseGroups * kFuseLanes);
    e = is_max ? launch(k_fuse_unions<true>, g, blk, 0, s, fa)
               : launch(k_fuse_unions<false>, g, blk, 0, s, fa);
    if (e != hipSuccess) return e;
  }
  note_split(plan.P, 1, 1, 2);
  return hipSuccess;
}

// k_strip_prepare behind the copy of a batch's camera state: geometry and row tables of every frame.
hipError_t launch_prepare(const dm_params& p, const dm_frames_plan& fp, const PreparedView& pv, hipStream_t s) {
  StripPrepArgs pa;
  pa.cfg = pv.cfg; pa.frames = pv.frames;
  pa.slab_stride = fp.slab_cells; pa.max_rows = fp.max_rows; pa.mw = p.mw; pa.list_cap = list_cap_of(p, fp);
  pa.t = pv.t; pa.status = pv.status;
  return launch(k_strip_prepare, dim3(p.B, kPrepSlices), dim3(kPrepThreads), 0, s, pa);
}

dm_frames_plan plan_of(const Plan& plan, const Rig& rg) {
  dm_frames_plan fp;
  memset(&fp, 0, sizeof(fp));
  fp.strips = plan.P; fp.strip_width = plan.wp; fp.slab_cells = rg.slab_stride;
  fp.max_rows = rg.max_rows; fp.max_union_cells = rg.max_union; fp.slack_cells = rg.slack;
  return fp;
}

}  // namespace

// hipErrorNotSupported: the strip path does not apply to this call (nothing enqueued).
hipError_t run_strip(const dm_params& p, const dm_frame* frames_host, const float* depth,
                     const float* value, const uint8_t* valid, float* out, uint8_t* mask,
                     float* height, float* fused, uint8_t* fused_mask, void* ws, size_t ws_bytes,
                     hipEvent_t before_projection, hipEvent_t after_projection, hipStream_t s) {
  if (g_force_legacy || p.B > 65535) return hipErrorNotSupported;
  if (!aligned_for_strips(depth, value, valid, out, mask, height, fused, fused_mask)) return hipErrorNotSupported;
  const Rig* rg = nullptr;
  const Plan* plan = plan_and_rig(p, frames_host, &rg);
  if (!plan) return hipErrorNotSupported;
  const size_t head = prepared_bytes(p);
  if (reinterpret_cast<uintptr_t>(ws) % 256 != 0 || ws_bytes < head) return hipErrorNotSupported;
  const dm_frames_plan fp = plan_of(*plan, *rg);
  {   // would the rest fit?  (nothing may be enqueued before the answer is yes)
    Layout l;
    if (!carve(p, static_cast<unsigned char*>(ws) + head, ws_bytes - head, l)) return hipErrorNotSupported;
    const size_t hm = (height && value) ? up256((size_t)p.B * p.dc * p.mh * p.mw) : 0;
    if (l.slab_bytes < hm + (size_t)p.B * plan->P * fp.slab_cells * 4) return hipErrorNotSupported;
  }
  hipError_t e = hipSuccess;
  if (before_projection) {
    e = hipEventRecord(before_projection, s);
    if (e != hipSuccess) return e;
  }
  // one stream-ordered copy (the runtime has copied pageable memory out by the time it returns)
  const std::vector<unsigned char>& stage = stage_prepared(p, *rg, frames_host);
  e = hipMemcpyAsync(ws, stage.data(), stage.size(), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  e = launch_prepare(p, fp, view_prepared(p, ws), s);
  if (e != hipSuccess) return e;
  return launch_strips(p, *plan, fp, view_prepared(p, ws), depth, value, valid, out, mask, height, fused,
                       fused_mask, static_cast<unsigned char*>(ws) + head, ws_bytes - head, after_projection, s);
}

size_t strip_prepared_bytes(const dm_params& p) { return prepare_plan(p) ? prepared_bytes(p) : 0; }

// dm_frames_prepare_f32: validate the batch's camera state, size the launches, upload, and derive
// the frames' geometry and row tables on the device (k_strip_prepare).
hipError_t strip_prepare(const dm_params& p, const dm_frame* frames_host, void* prepared_dev,
                         size_t prepared_size, dm_frames_plan* plan_out, hipStream_t s) {
  if (g_force_legacy || p.B > 65535 || p.B < 1) return hipErrorNotSupported;
  const Rig* rg = nullptr;
  const Plan* plan = plan_and_rig(p, frames_host, &rg, true);
  if (!plan) return hipErrorNotSupported;
  if (reinterpret_cast<uintptr_t>(prepared_dev) % 256 != 0 || prepared_size < prepared_bytes(p))
    return hipErrorInvalidValue;
  *plan_out = plan_of(*plan, *rg);
  const std::vector<unsigned char>& stage = stage_prepared(p, *rg, frames_host);
  const hipError_t e = hipMemcpyAsync(prepared_dev, stage.data(), stage.size(), hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return e;
  return launch_prepare(p, *plan_out, view_prepared(p, prepared_dev), s);
}

// dm_orth_project_prepared_f32
hipError_t run_strip_prepared(const dm_params& p, const dm_frames_plan& fp, void* prepared_dev,
                              const float* depth, const float* value, const uint8_t* valid, float* out,
                              uint8_t* mask, float* height, float* fused, uint8_t* fused_mask, void* ws,
                              size_t ws_bytes, hipEvent_t before_projection, hipEvent_t after_projection,
                              hipStream_t s) {
  if (!aligned_for_strips(depth, value, valid, out, mask, height, fused, fused_mask)) return hipErrorNotSupported;
  const Plan* plan = cached_plan(p);
  if (!plan || plan->P != fp.strips) plan = cached_plan(p, fp.strips);
  if (!plan || plan->P != fp.strips || plan->wp != fp.strip_width) return hipErrorNotSupported;
  if (before_projection) {
    const hipError_t e = hipEventRecord(before_projection, s);
    if (e != hipSuccess) return e;
  }
  return launch_strips(p, *plan, fp, view_prepared(p, prepared_dev), depth, value, valid, out, mask, height,
                       fused, fused_mask, ws, ws_bytes, after_projection, s);
}

}  // namespace dm

extern "C" __attribute__((visibility("default"))) int dm_debug_force_strips(int strips) {
  const int old = dm::g_force_strips;
  dm::g_force_strips = strips > 0 && strips <= dm::strip::kMaxStrips ? strips : 0;
  return old;
}

extern "C" __attribute__((visibility("default"))) int dm_debug_strip_value_list(int on) {
  const int old = !dm::g_no_value_list;
  dm::g_no_value_list = on == 0;
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
  compute_rig(*p, plan, frames[0], magnitude < (1 << 18) ? (1 << 18) : magnitude + (1 << 17), rg);
  if (out_bound) {
    out_bound[0] = rg.slack; out_bound[1] = rg.fits; out_bound[2] = rg.slab_stride;
    out_bound[3] = rg.max_rows; out_bound[4] = rg.max_union;
  }
  const int stride = 8 + 4 * strip::kMaxStrips;
  for (int b = 0; b < p->B; ++b) {
    dm_frame f = frames[b];
    if (!p->to_global) {      // as staged for the device: neutral yaw, no translation
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

// GPU: the same geometry from k_strip_geometry_dump (the device's lane-parallel evaluation),
// for the test that host and device agree bit for bit.  frames_dev (B, 32) f32 (a local map's
// records with a neutral yaw), frames_host: the same records on the host (for the rig);
// geom_dev: B * sizeof(FrameGeom) + 1024 bytes of device scratch.
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
  compute_rig(*p, plan, frames_host[0], magnitude < (1 << 18) ? (1 << 18) : magnitude + (1 << 17), rg);
  const size_t need = (size_t)p->B * sizeof(strip::FrameGeom) + kCfgBytes;
  if (geom_bytes < need) return -(int)need;
  strip::Cfg* cfg_dev = reinterpret_cast<strip::Cfg*>(static_cast<unsigned char*>(geom_dev) +
                                                      (size_t)p->B * sizeof(strip::FrameGeom));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(cfg_dev, &rg.cfg, sizeof(strip::Cfg), hipMemcpyHostToDevice, s) != hipSuccess) return -2;
  if (hipStreamSynchronize(s) != hipSuccess) return -2;       // (rg is a local)
  hipLaunchKernelGGL(k_strip_geometry_dump, dim3(p->B), dim3(64), 0, s, cfg_dev, frames_dev,
                     static_cast<strip::FrameGeom*>(geom_dev));
  return hipGetLastError() == hipSuccess ? plan.P : -2;
}

#ifdef DM_STAMPS
extern "C" __attribute__((visibility("default"))) void dm_debug_strip_stamp_buffer(long long* dev) {
  dm::g_stamp_buffer = dev;
}
#endif
