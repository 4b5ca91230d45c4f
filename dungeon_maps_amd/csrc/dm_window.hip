// LDS-windowed fast path of orth_project (max / min, one value per pixel).
//
// Why: the scatter has no locality the memory system can exploit -- global
// atomics run at ~27 G/s on MI355X whatever their scope (profiles/r01_microbench.log),
// LDS atomics at ~5300 G/s.  So every reduction happens in LDS and each output
// byte is written exactly once, by plain coalesced stores.
//
//   k_window_scatter  one workgroup per (frame, channel, image part).  A part
//       is a column strip (x row band) of the depth image.  The host bounds the
//       part's footprint in the map -- the frustum slab of its pixel rectangle
//       between trunc_depth_min and trunc_depth_max is a convex polytope whose
//       extreme cells are reached at its 8 corners -- and the workgroup
//       accumulates into an LDS image of exactly that window with ds_max_f32 /
//       ds_min_f32, then flushes the window as a "slab" (plain 16-byte stores,
//       read back by the next kernel).  Interleaved with the projection it also
//       writes its share of the map rows outside the frame's union window
//       (fill value, mask 0): ~77 % of the output never sees a second kernel.
//   k_window_merge    one thread per 4 cells of the union window: combines the
//       <= P slabs covering them (windows of neighbouring strips overlap) and
//       writes map and mask once.
//   k_fuse_unions / k_fuse_windows  batch fuse (max / min over the frames), from
//       the finished maps or straight from the slabs.
//
// Everything a workgroup needs is wave-uniform (frame record, windows: one staged
// table per launch) and read through scalar loads; depth is read with 16-byte
// loads, one row segment per group of lanes, several rows in flight per thread,
// the first of them requested before anything but the kernel arguments is known.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cstddef>
#include <type_traits>
#include <vector>

#include "dm_kernels.hpp"
#include "dm_window_geometry.hpp"
#include "dm_window_kernels.hpp"

namespace dm {

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

bool bounded_depth_params(dm_params& p, const dm_frame* frames_host) { return bound_depth_range(p, frames_host); }

bool window_path_supported(const dm_params& p) {
  if (p.reduction == DM_REDUCE_PROD) return false;      // generic path
  if (p.mw % 4 != 0) return false;
  if (p.mw > 32767 || p.mh > 32767) return false;   // Win16
  if ((int64_t)p.mh * p.mw >= (1ll << 28)) return false;   // (32-bit byte offsets of the fill duty's buffer stores)
  if (!(p.fill == p.fill)) return false;       // NaN fill has no order
  return true;
}

// Device copies of the call's geometry, written by k_window_scatter for the kernels that
// follow it: part windows, union windows.
static int frames_per_chunk(int nparts) {
  const int win_stride = nparts <= kFewParts ? kFewParts : nparts;
  const int chunk = kChunkWins / win_stride;
  return chunk > kChunkFrames ? kChunkFrames : chunk;
}

static size_t geometry_bytes(int B, int nparts) {
  const int chunk = frames_per_chunk(nparts);
  return align_up((size_t)B * nparts * sizeof(Win16), 256) + align_up((size_t)B * sizeof(Win16), 256) +
         align_up((size_t)((B + chunk - 1) / chunk) * sizeof(ScatterTables), 256) +
         align_up((size_t)B * sizeof(FlowRec), 256);       // (fused ego-motion flow: dm_orth_project_flow_f32)
}

static constexpr size_t kSlabBudget = (size_t)256 << 20;   // slab bytes per channel group

size_t window_workspace_bytes(const dm_params& p) {
  // slabs of one channel group (+ the scratch mask of the height pass).  Sized for up
  // to 4x the default number of parts (run_window splits further only when windows do
  // not fit in LDS, and falls back to the generic path if the workspace cannot hold that).
  size_t cap = (size_t)p.mh * p.mw * (p.reduction == DM_REDUCE_MEAN ? 2 : 1);
  if (cap > kMaxLdsBytes / 4) cap = kMaxLdsBytes / 4;
  const Parts d = choose_parts(p, 1);
  size_t np = (size_t)d.pc * d.pr * 4;
  if (np > kMaxParts) np = kMaxParts;
  const size_t oc = p.vc ? p.vc : p.dc;
  const size_t one = (size_t)p.B * np * align_up(cap, 4) * 4;     // one channel of every frame
  size_t slabs = one * oc;
  if (slabs > kSlabBudget) slabs = one > kSlabBudget ? one : kSlabBudget;
  const size_t height_mask = p.vc ? align_up((size_t)p.B * p.dc * p.mh * p.mw, 256) : 0;
  return geometry_bytes(p.B, kMaxParts) + slabs + height_mask;
}

namespace {

thread_local size_t g_slab_budget = 0;             // dm_debug_slab_budget (0: no cap)

struct Staged {                 // what run_window keeps between its passes
  Parts parts;
  int nparts, slab_stride, max_union;
  int max_tiles;                // k_window_merge_tiled blocks of the largest union window
  int gx0, gz0, gx1, gz1;       // bounding box of all union windows (empty: gx1 <= gx0)
  Win16* g_wins;                // device copies (workspace head)
  Win16* g_unions;
  const ScatterTables* d_tables;  // one per chunk of frames: workspace, filled by ONE stream-ordered copy
  FlowRec* d_flow;              // (B) flow records, behind the tables (written only by a fused flow call)
  int chunk;                    // frames per chunk
  size_t geom_bytes;
  bool fast, fast_div;
  float res_inv, fx_inv, fy_inv;
  const FrameRec* frames;       // (B)            host, thread-local storage
  const Win16* wins;            // (B, nparts)
  const Win16* unions;          // (B)
};

template <class K, class... Args>
inline hipError_t launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t s,
                         const Args&... args) {
  hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
  return hipGetLastError();
}

using Kernel = void (*)(ScatterArgs, const ScatterTables*);

Kernel pick_kernel(int red, bool fast, bool has_valid, bool has_value, bool vec4, bool lean, bool stream = false) {
#define DM_K(M, F, V, S) {k_window_scatter<M, F, V, S, 1, false>, k_window_scatter<M, F, V, S, 4, false>}
#define DM_RED(M)                                                                  \
      {{{DM_K(M, false, false, false), DM_K(M, false, false, true)},                \
        {DM_K(M, false, true, false), DM_K(M, false, true, true)}},                 \
       {{DM_K(M, true, false, false), DM_K(M, true, false, true)},                  \
        {DM_K(M, true, true, false), DM_K(M, true, true, true)}}}
  // [kMin | kMax | kSum | kMean][fast][has_valid][has_value][vec4]
  static const Kernel table[4][2][2][2][2] = {DM_RED(kMin), DM_RED(kMax), DM_RED(kSum), DM_RED(kMean)};
#undef DM_RED
#undef DM_K
  static const Kernel lean_table[4][2] = {
      {k_window_scatter<kMin, true, false, false, 4, true>,
       k_window_scatter<kMin, true, false, true, 4, true>},
      {k_window_scatter<kMax, true, false, false, 4, true>,
       k_window_scatter<kMax, true, false, true, 4, true>},
      {k_window_scatter<kSum, true, false, false, 4, true>,
       k_window_scatter<kSum, true, false, true, 4, true>},
      {k_window_scatter<kMean, true, false, false, 4, true>,
       k_window_scatter<kMean, true, false, true, 4, true>}};
  static const Kernel stream_table[4][2] = {
      {k_window_scatter<kMin, true, false, false, 4, true, false, true>,
       k_window_scatter<kMin, true, false, true, 4, true, false, true>},
      {k_window_scatter<kMax, true, false, false, 4, true, false, true>,
       k_window_scatter<kMax, true, false, true, 4, true, false, true>},
      {k_window_scatter<kSum, true, false, false, 4, true, false, true>,
       k_window_scatter<kSum, true, false, true, 4, true, false, true>},
      {k_window_scatter<kMean, true, false, false, 4, true, false, true>,
       k_window_scatter<kMean, true, false, true, 4, true, false, true>}};
  if (lean && stream) return stream_table[red][has_value];
  return lean ? lean_table[red][has_value] : table[red][fast][has_valid][has_value][vec4];
}

Kernel pick_flow_kernel(bool is_max) {
  return is_max ? k_window_scatter<kMax, true, false, false, 4, true, true>
                : k_window_scatter<kMin, true, false, false, 4, true, true>;
}

thread_local int g_last_flow_fused = 0;       // dm_debug_last_flow_fused
// dm_debug_flow_fused: whether dm_orth_project_flow_f32 lets the projection kernel compute the flow.
// Measured at BASELINE configs[4]'s frames (16 x 1280x960 -> 2048x2048, working set beyond the
// Infinity Cache): 213.6 us fused against 100.8 + 62.0 us for the two kernels -- the flow's two
// IEEE divisions per pixel are dependent chains, and the projection kernel has four waves per
// SIMD (its LDS window) to hide them where the stand-alone kernel has sixteen; the second depth
// read the fusion saves is 79 MB, about 13 us.  So the default is the two kernels.
thread_local int g_flow_fused = 0;

// One pass: scatter `value` (or the heights when NULL) of channels [0, oc_total)
// into out/mask, channel group by channel group, chunk of frames by chunk of frames.
// With `fused` set, the per-frame maps are skipped (out/mask NULL) and every channel
// group's slabs are reduced straight into the (oc_total, mh, mw) fused map.
hipError_t window_pass(const dm_params& p, const Staged& st, float* slabs,
                       const float* depth, const float* value, const uint8_t* valid, float* out,
                       uint8_t* mask, int oc_total, float fill, int red, size_t slab_bytes,
                       hipStream_t s, float* fused = nullptr, uint8_t* fused_mask = nullptr,
                       int accumulate = 0, float* flow_grid = nullptr, bool* flow_done = nullptr,
                       bool depth_read_again = false) {
  ScatterArgs sa;
  sa.flow = nullptr; sa.flow_grid = nullptr;
  sa.W = p.W; sa.H = p.H;
  sa.clip = p.clip_border > 0 ? p.clip_border : 0;
  sa.flip_h = p.flip_h != 0;
  sa.cx = p.cx; sa.cy = p.cy; sa.fx = p.fx; sa.fy = p.fy; sa.res = p.res;
  sa.res_inv = st.res_inv; sa.fx_inv = st.fx_inv; sa.fy_inv = st.fy_inv;
  sa.dmin = p.has_dmin ? p.dmin : -INFINITY;
  sa.dmax = p.has_dmax ? p.dmax : INFINITY;
  sa.hmax = p.has_hmax ? p.hmax : INFINITY;
  sa.Hm1 = (float)(p.H - 1); sa.mhm1 = (float)(p.mh - 1);
  sa.parts = st.parts;
  sa.dc = p.dc; sa.valid_c = p.valid_c;
  sa.oc_total = oc_total;
  sa.slab_stride = st.slab_stride;
  sa.fill = fill;
  sa.depth = depth; sa.value = value; sa.valid = valid;
  sa.slabs = slabs;
  sa.out = out; sa.mask = mask; sa.mh = p.mh; sa.mw = p.mw;

  const bool has_valid = valid != nullptr, has_value = value != nullptr;
  const bool vec4 = (p.W % 4 == 0) && (reinterpret_cast<uintptr_t>(depth) % 16 == 0) &&
                    (!value || reinterpret_cast<uintptr_t>(value) % 16 == 0) &&
                    (st.parts.wp % 4 == 0);
  sa.g_wins = st.g_wins; sa.g_unions = st.g_unions;
#ifdef DM_STAMPS
  sa.stamps = g_stamp_buffer;
#endif
  const size_t lds_bytes = align_up((size_t)st.slab_stride * 4, 16) + 64 * 4;   // + dummy cells
  // lean variant: both depth bounds finite, no height truncation, no border, no valid map
  const bool lean = st.fast && vec4 && !has_valid && p.has_dmin && p.has_dmax &&
                    isfinite(p.dmin) && isfinite(p.dmax) && !p.has_hmax && p.clip_border <= 0;
  const bool is_max = red == kMax;             // (the fuse kernels: max / min only)
  // the ego-motion flow rides on the lean height projection (every pixel is visited exactly once
  // per launch: no depth bands, one depth channel group)
  bool with_flow = flow_grid != nullptr && g_flow_fused && lean && !has_value && (red == kMax || red == kMin) &&
                   st.parts.pd == 1 && reinterpret_cast<uintptr_t>(flow_grid) % 16 == 0;
  // (a part whose window is empty -- its pixels cannot reach the map -- leaves the kernel in front of the
  // pixel loop, and with it the flow of its pixels: such a call takes the stand-alone kernel.  Found by the
  // parity campaign's flow mode, round 4.)
  for (size_t i = 0; with_flow && i < (size_t)p.B * st.nparts; ++i)
    with_flow = (int)st.wins[i].w * st.wins[i].h > 0;
  if (flow_done) *flow_done = with_flow;
  if (with_flow) { sa.flow = st.d_flow; sa.flow_grid = flow_grid; }
  // the depth maps non-temporally: where what the call reads and writes once cannot stay cache resident
  // (more than half of the Infinity Cache) and no second reader of the depth maps follows inside the call
  // (the ego-motion flow kernel of dm_orth_project_flow_f32 finds them in the cache only under the default
  // policy: cfg5 flow call 153 -> 176 us; plain projection 101 -> 96 us)
  const size_t once = (size_t)p.B * p.dc * p.H * p.W * 4 + (size_t)p.B * oc_total * p.mh * p.mw * 5;
  const bool stream = !depth_read_again && once > ((size_t)128 << 20) && st.parts.pd == 1;
  const Kernel kfn = with_flow ? pick_flow_kernel(is_max)
                               : pick_kernel(red, st.fast, has_valid, has_value, vec4, lean, stream);
  hipError_t e = hipSuccess;
  {   // raise the dynamic-LDS limit once per kernel variant and device
    static thread_local const void* done[160][8] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const void* key = reinterpret_cast<const void*>(kfn);
    bool seen = false;
    int free_slot = -1;
    if (dev >= 0 && dev < 8) {
      for (int i = 0; i < 160; ++i) {
        if (done[i][dev] == key) { seen = true; break; }
        if (!done[i][dev] && free_slot < 0) free_slot = i;
      }
    }
    if (!seen) {
      e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsBytes);
      if (e != hipSuccess) return e;
      if (dev >= 0 && dev < 8 && free_slot >= 0) done[free_slot][dev] = key;
    }
  }

  const int chunk = st.chunk;                // frames per launch: what one table holds
  // channel groups: the slabs of one group fit the workspace's slab region
  const size_t per_channel = (size_t)p.B * st.nparts * st.slab_stride * 4;
  if (g_slab_budget && slab_bytes > g_slab_budget) slab_bytes = g_slab_budget;
  int group = (int)(slab_bytes / (per_channel ? per_channel : 1));
  if (group < 1) group = 1;
  if (group > oc_total) group = oc_total;
  if (group > 65535 / chunk) group = 65535 / chunk;       // merge grid.y = frames * channels
  for (int ch0 = 0; ch0 < oc_total; ch0 += group) {
    const int oc = oc_total - ch0 < group ? oc_total - ch0 : group;
    sa.oc = oc; sa.ch0 = ch0;
    for (int b0 = 0; b0 < p.B; b0 += chunk) {
      const int nb = p.B - b0 < chunk ? p.B - b0 : chunk;
      sa.b0 = b0;
      e = launch(kfn, dim3(st.nparts, oc, nb), dim3(kScatterThreads), lds_bytes, s, sa,
                 st.d_tables + b0 / chunk);
      if (e != hipSuccess) return e;
      if (!fused && st.max_union > 0) {
        MergeArgs ma;
        ma.b0 = b0; ma.oc = oc; ma.ch0 = ch0; ma.oc_total = oc_total; ma.mh = p.mh; ma.mw = p.mw;
        ma.nparts = st.nparts; ma.slab_stride = st.slab_stride; ma.fill = fill;
        {
          const ScatterTables* t = st.d_tables + b0 / chunk;     // device address arithmetic only
          ma.wins = t->wins; ma.unions = t->unions;
          ma.win_stride = st.nparts <= kFewParts ? kFewParts : st.nparts;
        }
        ma.slabs = slabs; ma.out = out; ma.mask = mask;
        if (st.nparts >= kTiledMergeParts) {
          const dim3 g((unsigned)st.max_tiles, nb * oc);
          e = red == kMax   ? launch(k_window_merge_tiled<kMax>, g, dim3(kMergeThreads), 0, s, ma)
              : red == kMin ? launch(k_window_merge_tiled<kMin>, g, dim3(kMergeThreads), 0, s, ma)
              : red == kSum ? launch(k_window_merge_tiled<kSum>, g, dim3(kMergeThreads), 0, s, ma)
                            : launch(k_window_merge_tiled<kMean>, g, dim3(kMergeThreads), 0, s, ma);
        } else {
          const dim3 g((unsigned)((st.max_union / 4 + kMergeThreads - 1) / kMergeThreads), nb * oc);
          e = red == kMax   ? launch(k_window_merge<kMax>, g, dim3(kMergeThreads), 0, s, ma)
              : red == kMin ? launch(k_window_merge<kMin>, g, dim3(kMergeThreads), 0, s, ma)
              : red == kSum ? launch(k_window_merge<kSum>, g, dim3(kMergeThreads), 0, s, ma)
                            : launch(k_window_merge<kMean>, g, dim3(kMergeThreads), 0, s, ma);
        }
        if (e != hipSuccess) return e;
      }
    }
    if (fused) {     // every frame's slabs of this channel group are in place
      FuseWinArgs fa;
      fa.nwin = p.B * st.nparts; fa.b0 = 0; fa.nparts = st.nparts; fa.oc = oc; fa.ch0 = ch0;
      fa.oc_total = oc_total; fa.mh = p.mh; fa.mw = p.mw; fa.slab_stride = st.slab_stride;
      fa.accumulate = accumulate; fa.fill = fill;
      fa.gx0 = st.gx0; fa.gz0 = st.gz0; fa.gx1 = st.gx1; fa.gz1 = st.gz1;
      fa.wins = st.g_wins; fa.slabs = slabs; fa.fused = fused; fa.fused_mask = fused_mask;
      fa.spans = nullptr; fa.span_rows = 0;        // (whole window rows in the slabs)
      const int bw4 = (st.gx1 - st.gx0) / 4;
      const int heavy = st.gx1 > st.gx0 ? ((bw4 + kFuseGroups - 1) / kFuseGroups) * (st.gz1 - st.gz0) : 0;
      const int per_fill_block = kFuseGroups * kFuseLanes * 8;
      const int fill_blocks = (int)(((size_t)p.mh * p.mw / 4 + per_fill_block - 1) / per_fill_block);
      const dim3 g((unsigned)(heavy + fill_blocks), oc);
      const dim3 blk(kFuseGroups * kFuseLanes);
      e = is_max ? launch(k_fuse_windows<true>, g, blk, 0, s, fa)
                 : launch(k_fuse_windows<false>, g, blk, 0, s, fa);
      if (e != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

// What the previous calls of a shape settled on: the split into parts (stage_windows) and
// whether the frames have to go through in halves (run_window).  Per thread, a few shapes.
struct Remembered {
  dm_params key;
  Parts parts;
  bool valid, halve;
  int hopeless;          // calls left before the search is tried again after it failed
};
Remembered& remembered(const dm_params& p) {
  thread_local Remembered slots[4] = {};
  thread_local int next = 0;
  for (Remembered& r : slots)
    if (memcmp(&r.key, &p, sizeof(dm_params)) == 0) return r;
  Remembered& r = slots[next];
  next = (next + 1) % 4;
  r = Remembered{};
  r.key = p;
  return r;
}

thread_local int g_last_split[4] = {0, 0, 0, 0};   // dm_debug_last_split
thread_local int g_last_path = 0;                  // dm_debug_last_path
thread_local bool g_force_bands = false;           // dm_debug_force_bands

// Depth bands multiply the workgroups, each with a window to initialise and flush: with few
// pixels per part that costs more than the generic path's atomics.  Both sides as measured on
// MI355X: generic = 20 ps per point + fill at 5.5 TB/s; windowed = per pass
// waves of 256 workgroups at 4 us + (slab + part pixels) at 20 GB/s per CU, plus the merge
// reading the slabs at 3.7 TB/s.
bool banded_split_pays(const dm_params& p, const Parts& parts, int nparts, int max_area,
                       size_t slab_capacity) {
  const double oc = p.vc ? p.vc : p.dc;
  const double points = (double)p.B * p.H * p.W * oc, cells = (double)p.B * oc * p.mh * p.mw;
  const double t_generic = points * 20e-6 + cells * 5.0 / 5.5e6;
  const double slab = (double)align_up((size_t)max_area, 4) * 4.0;
  double frames = p.B;
  int passes = 1;
  while (frames > 1.0 && frames * nparts * slab > (double)slab_capacity) {
    frames = ceil(frames / 2.0);
    passes *= 2;
  }
  const double t_wg = 4.0 + (slab + (double)parts.wp * parts.hp * 4.0) / 20480.0;
  const double waves = ceil(frames * oc * nparts / 256.0);
  const double t_window = passes * (waves * t_wg + 6.0) + (double)p.B * oc * nparts * slab / 3.7e6;
  return t_window < t_generic;
}

// Host-side geometry of a call: parts, part windows, frame records.  Returns
// hipErrorNotSupported when the windows cannot be made to fit in LDS (the caller then
// takes the generic path); nothing has been enqueued in that case.
hipError_t stage_windows(const dm_params& p, const dm_frame* frames_host, void* ws,
                         size_t ws_bytes, Staged& st, size_t& slab_bytes, hipStream_t s,
                         hipEvent_t before = nullptr) {
  if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return hipErrorNotSupported;
  thread_local std::vector<FrameRec> recs;
  thread_local std::vector<Win16> wins;      // (B, nparts) then (B) unions
  // more, narrower parts until every window fits in LDS; when splitting the image does not
  // get there (a long thin wedge: fine resolution, long range) the depth range is split too
  int max_area = 0;
  const int planes = p.reduction == DM_REDUCE_MEAN ? 2 : 1;     // mean: sum window + count window
  const bool can_band = p.has_dmin && p.has_dmax && isfinite(p.dmin) && p.dmax > p.dmin &&
                        isfinite(p.dmax);
  thread_local std::vector<PartSlopes> slopes;
  const bool bounded = frustum_bounded(p);
  auto evaluate = [&](const Parts& parts) -> bool {
    const int pd = parts.pd;
    st.parts = parts;
    const int image_parts = parts.pc * parts.pr;
    st.nparts = image_parts * pd;
    wins.resize((size_t)p.B * (st.nparts + 1));
    Win16* unions = wins.data() + (size_t)p.B * st.nparts;
    slopes.resize(image_parts);
    for (int pr = 0; pr < parts.pr; ++pr)
      for (int pc = 0; pc < parts.pc; ++pc) {
        const int q0 = pc * parts.wp, r0 = pr * parts.hp;
        const int q1 = q0 + parts.wp < p.W ? q0 + parts.wp : p.W;
        const int r1 = r0 + parts.hp < p.H ? r0 + parts.hp : p.H;
        slopes[pr * parts.pc + pc] = part_slopes(p, q0, q1, r0, r1);
      }
    max_area = 0; st.max_union = 0; st.max_tiles = 0;
    st.gx0 = p.mw; st.gz0 = p.mh; st.gx1 = 0; st.gz1 = 0;
    for (int b = 0; b < p.B; ++b) {
      int ux0 = p.mw, ux1 = 0, uz0 = p.mh, uz1 = 0;
      const FrameAffine fa = frame_affine(p, frames_host[b]);
      Win16* row = wins.data() + (size_t)b * st.nparts;
      for (int k = 0; k < pd; ++k) {
        float dlo = p.dmin, dhi = p.dmax;
        if (pd > 1) band_bounds(p.dmin, p.dmax, pd, k, dlo, dhi, parts.geo);
        for (int ip = 0; ip < image_parts; ++ip) {
          const Window w = part_window(p, fa, slopes[ip], bounded, dlo, dhi);
          row[k * image_parts + ip] = narrow(w);
          if (w.w * w.h > max_area) max_area = w.w * w.h;
          if (w.w > 0) {
            if (w.x0 < ux0) ux0 = w.x0;
            if (w.x0 + w.w > ux1) ux1 = w.x0 + w.w;
            if (w.z0 < uz0) uz0 = w.z0;
            if (w.z0 + w.h > uz1) uz1 = w.z0 + w.h;
          }
        }
      }
      const Window U = ux1 > ux0 ? Window{ux0, uz0, ux1 - ux0, uz1 - uz0} : Window{0, 0, 0, 0};
      unions[b] = narrow(U);
      if (U.w * U.h > st.max_union) st.max_union = U.w * U.h;
      const int tiles = ((U.w / 4 + kTileGroups - 1) / kTileGroups) * ((U.h + kTileRows - 1) / kTileRows);
      if (tiles > st.max_tiles) st.max_tiles = tiles;
      if (U.w > 0) {
        if (U.x0 < st.gx0) st.gx0 = U.x0;
        if (U.z0 < st.gz0) st.gz0 = U.z0;
        if (U.x0 + U.w > st.gx1) st.gx1 = U.x0 + U.w;
        if (U.z0 + U.h > st.gz1) st.gz1 = U.z0 + U.h;
      }
    }
    static const bool verbose = getenv("DM_DEBUG_WINDOWS") != nullptr;
    if (verbose)
      fprintf(stderr, "[dm] pd=%d parts=%dx%d nparts=%d max_area=%d max_union=%d\n", pd, parts.pc,
              parts.pr, st.nparts, max_area, st.max_union);
    return (size_t)max_area * 4 * planes + 64 * 4 + 16 <= (size_t)kMaxLdsBytes;
  };
  // the split that worked for the previous call of the same shape is tried first (the answer
  // moves with the poses only): one evaluation per call in the steady state
  Remembered& last = remembered(p);
  if (last.hopeless > 0 && !g_force_bands) {
    --last.hopeless;
    g_last_split[0] = g_last_split[1] = g_last_split[2] = 0;
    g_last_path = 0;
    return hipErrorNotSupported;
  }
  const bool same_shape = last.valid && (last.parts.pd == 1 || can_band);
  bool fits = same_shape && evaluate(last.parts);
  if (!fits) {
    // candidates: the image splits without bands first (fewest parts first), then the banded
    // ones by their number of parts
    std::vector<Parts> cands;
    auto seen = [&](const Parts& c) {
      for (const Parts& o : cands)
        if (o.pc == c.pc && o.pr == c.pr && o.pd == c.pd) return true;
      return false;
    };
    for (int pd = 1; pd <= 8; pd *= 2) {
      if (pd > 1 && !can_band) break;
      for (int min_parts = 1;; min_parts *= 2) {
        const Parts c = choose_parts(p, min_parts, pd);
        const int image_parts = c.pc * c.pr;
        if (image_parts * pd > kMaxParts || image_parts < min_parts) break;   // cannot split further
        if (!seen(c)) cands.push_back(c);
        // a window that is the whole map cannot shrink by splitting the image
        if (!p.has_dmin || !p.has_dmax || image_parts >= 64) break;
      }
    }
    std::stable_sort(cands.begin(), cands.end(), [](const Parts& x, const Parts& y) {
      if ((x.pd == 1) != (y.pd == 1)) return x.pd == 1;
      return x.pc * x.pr * x.pd < y.pc * y.pr * y.pd;
    });
    for (const Parts& c0 : cands) {
      if (same_shape && c0.pc == last.parts.pc && c0.pr == last.parts.pr && c0.pd == last.parts.pd) continue;
      // (banded splits of a range that reaches beyond the map: geometric band edges first, then
      // equal steps)
      bool done = false;
      for (int geo = c0.geo; geo >= 0 && !done; --geo) {
        Parts c = c0;
        c.geo = geo;
        if (evaluate(c)) {
          const size_t geom = geometry_bytes(p.B, st.nparts);
          fits = c.pd == 1 || g_force_bands ||
                 banded_split_pays(p, c, st.nparts, max_area * planes, ws_bytes > geom ? ws_bytes - geom : 0);
          done = true;    // splits of more parts cost more still
        }
      }
      if (done) break;
    }
  }
  last.valid = fits;
  if (fits) last.parts = st.parts;
  else last.hopeless = 63;       // the search is host time: not on every call of such a shape
  if (!fits) {
    g_last_split[0] = g_last_split[1] = g_last_split[2] = 0;
    g_last_path = 0;
    return hipErrorNotSupported;
  }
  g_last_split[0] = st.parts.pc; g_last_split[1] = st.parts.pr; g_last_split[2] = st.parts.pd;
  g_last_path = 1;
  st.slab_stride = (int)align_up((size_t)(max_area > 0 ? max_area : 4), 4) * planes;
  st.geom_bytes = geometry_bytes(p.B, st.nparts);
  if (ws_bytes < st.geom_bytes + (size_t)p.B * st.nparts * st.slab_stride * 4)
    return hipErrorOutOfMemory;       // run_window then takes the frames in two halves
  slab_bytes = ws_bytes - st.geom_bytes;
  {
    unsigned char* base = static_cast<unsigned char*>(ws);
    st.g_wins = reinterpret_cast<Win16*>(base);
    base += align_up((size_t)p.B * st.nparts * sizeof(Win16), 256);
    st.g_unions = reinterpret_cast<Win16*>(base);
    base += align_up((size_t)p.B * sizeof(Win16), 256);
    st.d_tables = reinterpret_cast<const ScatterTables*>(base);
    const int chunk_now = frames_per_chunk(st.nparts);
    base += align_up((size_t)((p.B + chunk_now - 1) / chunk_now) * sizeof(ScatterTables), 256);
    st.d_flow = reinterpret_cast<FlowRec*>(base);
  }

  recs.resize(p.B);
  for (int b = 0; b < p.B; ++b) {
    const dm_frame& f = frames_host[b];
    FrameRec& r = recs[b];
    memcpy(r.p, f.Rp, sizeof(r.p));
    r.cam_h = f.cam_height;
    r.wo = f.width_offset; r.ho = f.height_offset; r.pad = 0.0f;
    if (p.to_global) {
      memcpy(r.y, f.Ry, sizeof(r.y));
      r.tx = f.tx; r.tz = f.tz;
    } else {              // local map: neutral yaw, no translation (exact: x*1 + z*0 + 0)
      static const float eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      memcpy(r.y, eye, sizeof(eye));
      r.tx = 0.0f; r.tz = 0.0f;
    }
  }
  // branch-free exact division needs exactly rounded reciprocals and sane magnitudes
  st.fast_div = exact_reciprocal(p.res, &st.res_inv) && exact_reciprocal(p.fx, &st.fx_inv) &&
                exact_reciprocal(p.fy, &st.fy_inv) && p.res >= 1e-6f && p.res <= 1e6f &&
                p.fx >= 1e-6f && p.fx <= 1e6f && p.fy >= 1e-6f && p.fy <= 1e6f;
  if (!st.fast_div) st.res_inv = st.fx_inv = st.fy_inv = 0.0f;
  st.fast = st.fast_div && axis_aligned(recs.data(), p.B);
  st.frames = recs.data();
  st.wins = wins.data();
  st.unions = wins.data() + (size_t)p.B * st.nparts;
  // the launches' tables (thread-local staging: hipMemcpyAsync from pageable memory has
  // copied the bytes out by the time it returns)
  thread_local std::vector<ScatterTables> tabs;
  st.chunk = frames_per_chunk(st.nparts);
  const int win_stride = st.nparts <= kFewParts ? kFewParts : st.nparts;
  const int nchunks = (p.B + st.chunk - 1) / st.chunk;
  tabs.resize(nchunks);
  for (int c = 0; c < nchunks; ++c) {
    ScatterTables& tab = tabs[c];
    const int b0 = c * st.chunk;
    const int nb = p.B - b0 < st.chunk ? p.B - b0 : st.chunk;
    memcpy(tab.frames, st.frames + b0, (size_t)nb * sizeof(FrameRec));
    for (int i = 0; i < nb; ++i)
      memcpy(tab.wins + (size_t)i * win_stride, st.wins + (size_t)(b0 + i) * st.nparts,
             (size_t)st.nparts * sizeof(Win16));
    memcpy(tab.unions, st.unions + b0, (size_t)nb * sizeof(Win16));
  }
  if (before) {
    const hipError_t e = hipEventRecord(before, s);
    if (e != hipSuccess) return e;
  }
  const size_t table_bytes =
      nchunks > 1 ? (size_t)nchunks * sizeof(ScatterTables)
                  : offsetof(ScatterTables, wins) + (size_t)p.B * win_stride * sizeof(Win16);
  return hipMemcpyAsync(const_cast<ScatterTables*>(st.d_tables), tabs.data(), table_bytes,
                        hipMemcpyHostToDevice, s);
}

}  // namespace

hipError_t run_window(const dm_params& p, const dm_frame* frames_host, const float* depth,
                      const float* value, const uint8_t* valid, float* out, uint8_t* mask,
                      float* height, float* fused, uint8_t* fused_mask, void* ws,
                      size_t ws_bytes, hipEvent_t before_projection, hipEvent_t after_projection,
                      hipStream_t s, const dm_frame* flow_frames_host, float* flow_grid,
                      bool* flow_done) {
  const int oc_total = p.vc ? p.vc : p.dc;
  if (flow_done) *flow_done = false;
  g_last_flow_fused = 0;
  if (reinterpret_cast<uintptr_t>(out) % 16 != 0 || reinterpret_cast<uintptr_t>(mask) % 4 != 0 ||
      reinterpret_cast<uintptr_t>(fused) % 16 != 0 ||
      reinterpret_cast<uintptr_t>(fused_mask) % 4 != 0 ||
      reinterpret_cast<uintptr_t>(height) % 16 != 0)
    return hipErrorNotSupported;
  Staged st;
  size_t slab_bytes = 0;
#ifdef DM_HOST_TIMING
  using clk = std::chrono::steady_clock;
  thread_local double acc_t[4] = {0, 0, 0, 0};
  thread_local int calls_t = 0;
  const auto t_0 = clk::now();
#endif
  Remembered& shape = remembered(p);
  if (shape.hopeless > 0 && !g_force_bands) {
    --shape.hopeless;
    g_last_split[0] = g_last_split[1] = g_last_split[2] = 0;
    g_last_path = 0;
    return hipErrorNotSupported;
  }
  hipError_t e = shape.halve && !fused && p.B >= 2
                     ? hipErrorOutOfMemory
                     : stage_windows(p, frames_host, ws, ws_bytes, st, slab_bytes, s, before_projection);
  if (e == hipErrorOutOfMemory) {
    // many parts of large windows (fine resolution): the slabs of all frames do not fit the
    // workspace.  Frames are independent, so they go through in two halves.
    if (p.B < 2 || fused) return hipErrorNotSupported;
    shape.halve = true;
    g_last_split[3] += 1;
    dm_params q = p;
    q.B = p.B / 2;
    const size_t n = (size_t)p.H * p.W, m = (size_t)p.mh * p.mw, h = q.B;
    // (a fused ego-motion flow is not carried through the halves: the caller runs the stand-alone kernel)
    e = run_window(q, frames_host, depth, value, valid, out, mask, height, nullptr, nullptr, ws,
                   ws_bytes, before_projection, nullptr, s, nullptr, nullptr, nullptr);
    if (e == hipErrorNotSupported) shape.hopeless = 63;     // a half that does not fit or pay
    if (e != hipSuccess) return e;
    q.B = p.B - (int)h;
    e = run_window(q, frames_host + h, depth + h * p.dc * n, value ? value + h * p.vc * n : nullptr,
                      valid ? valid + h * p.valid_c * n : nullptr, out + h * oc_total * m,
                      mask + h * oc_total * m, height ? height + h * p.dc * m : nullptr, nullptr,
                      nullptr, ws, ws_bytes, nullptr, after_projection, s, nullptr, nullptr, nullptr);
    if (e == hipErrorNotSupported) shape.hopeless = 63;
    return e;
  }
  if (e != hipSuccess) return e;
#ifdef DM_HOST_TIMING
  const auto t_1 = clk::now();
#endif
  unsigned char* base = static_cast<unsigned char*>(ws);
  float* slabs = reinterpret_cast<float*>(base + st.geom_bytes);

  const bool is_max = p.reduction == DM_REDUCE_MAX;
  const int red = p.reduction == DM_REDUCE_MAX   ? kMax
                  : p.reduction == DM_REDUCE_MIN ? kMin
                  : p.reduction == DM_REDUCE_SUM ? kSum
                                                 : kMean;
  const bool flow_wanted = flow_grid != nullptr;       // (the stand-alone flow kernel reads the depth maps again)
  if (flow_grid && flow_frames_host && !value) {
    // the frames' flow records (rotation and translation of the pose transition, rotate(X, -pitch)):
    // one more stream-ordered copy, behind the tables
    thread_local std::vector<FlowRec> frecs;
    frecs.resize(p.B);
    for (int b = 0; b < p.B; ++b) {
      const dm_frame& f = flow_frames_host[b];
      FlowRec& r = frecs[b];
      memcpy(r.ry, f.Ry, sizeof(r.ry)); memcpy(r.ri, f.reserved, sizeof(r.ri));
      r.tx = f.tx; r.tz = f.tz; r.pad[0] = r.pad[1] = r.pad[2] = r.pad[3] = 0.0f;
    }
    e = hipMemcpyAsync(st.d_flow, frecs.data(), (size_t)p.B * sizeof(FlowRec), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
  } else {
    flow_grid = nullptr;
  }
  bool flowed = false;
  e = window_pass(p, st, slabs, depth, value, valid, out, mask, oc_total, p.fill, red,
                  slab_bytes, s, nullptr, nullptr, 0, flow_grid, &flowed,
                  /*depth_read_again=*/flow_wanted || (height && value));
  if (e != hipSuccess) return e;
  if (flow_done) *flow_done = flowed;
  g_last_flow_fused = flowed ? 1 : 0;
  if (height && value) {      // maps.py:332-350: second projection, NINF fill, max
    // its mask is not returned (maps.py:340): it goes to scratch at the workspace tail
    const size_t hm = (size_t)p.B * p.dc * p.mh * p.mw;
    if (slab_bytes < hm + (size_t)p.B * st.nparts * st.slab_stride * 4) return hipErrorNotSupported;
    uint8_t* scratch_mask = base + ws_bytes - hm;
    e = window_pass(p, st, slabs, depth, nullptr, valid, height, scratch_mask, p.dc, -INFINITY, kMax,
                    slab_bytes - hm, s);
    if (e != hipSuccess) return e;
  }
  if (after_projection) {
    e = hipEventRecord(after_projection, s);
    if (e != hipSuccess) return e;
  }
#ifdef DM_HOST_TIMING
  const auto t_2 = clk::now();
#endif
  if (fused) {
    FuseArgs fa;
    fa.B = p.B; fa.b0 = 0; fa.accumulate = 0;
    fa.dc = oc_total; fa.mh = p.mh; fa.mw = p.mw; fa.fill = p.fill;
    fa.unions = st.g_unions; fa.maps = out; fa.fused = fused; fa.fused_mask = fused_mask;
    const dim3 g((unsigned)(((size_t)p.mh * p.mw / 4 + kUnionGroups - 1) / kUnionGroups), oc_total);
    const dim3 blk(kUnionGroups * kUnionLanes);
    e = is_max ? launch(k_fuse_unions<true>, g, blk, 0, s, fa)
               : launch(k_fuse_unions<false>, g, blk, 0, s, fa);
    if (e != hipSuccess) return e;
  }
#ifdef DM_HOST_TIMING
  const auto t_3 = clk::now();
  acc_t[0] += std::chrono::duration<double, std::micro>(t_1 - t_0).count();
  acc_t[1] += std::chrono::duration<double, std::micro>(t_2 - t_1).count();
  acc_t[2] += std::chrono::duration<double, std::micro>(t_3 - t_2).count();
  if (++calls_t % 200 == 0) {
    fprintf(stderr, "[dm timing] stage %.2f us  window_pass %.2f us  fuse %.2f us per call\n",
            acc_t[0] / 200, acc_t[1] / 200, acc_t[2] / 200);
    acc_t[0] = acc_t[1] = acc_t[2] = 0;
  }
#endif
  return hipSuccess;
}

extern "C" __attribute__((visibility("default"))) int dm_debug_last_flow_fused(void) { return g_last_flow_fused; }
extern "C" __attribute__((visibility("default"))) int dm_debug_flow_fused(int on) {
  const int old = g_flow_fused;
  g_flow_fused = on != 0;
  return old;
}

// dm_orth_project_fused_f32: scatter into the LDS windows, then reduce the slabs of
// all frames straight into ONE (oc, mh, mw) map (optionally on top of its content).
hipError_t run_window_fused(const dm_params& p, const dm_frame* frames_host, const float* depth,
                            const float* value, const uint8_t* valid, float* out, uint8_t* mask,
                            int accumulate, void* ws, size_t ws_bytes, hipStream_t s) {
  if (reinterpret_cast<uintptr_t>(out) % 16 != 0 || reinterpret_cast<uintptr_t>(mask) % 4 != 0)
    return hipErrorNotSupported;
  Staged st;
  size_t slab_bytes = 0;
  hipError_t e = stage_windows(p, frames_host, ws, ws_bytes, st, slab_bytes, s, nullptr);
  if (e == hipErrorOutOfMemory) {
    // The slabs of all frames do not fit the workspace (many parts of large windows): the fuse is
    // a max / min, so the frames go through in two halves, the second one folding into the map
    // the first one left.  (Found by the fused parity campaign: B = 65 frames of 70x50 pixels
    // used to fail with "out of memory" here.)
    if (p.B < 2) return hipErrorNotSupported;
    dm_params q = p;
    q.B = p.B / 2;
    const size_t n = (size_t)p.H * p.W, h = q.B;
    e = run_window_fused(q, frames_host, depth, value, valid, out, mask, accumulate, ws, ws_bytes, s);
    if (e != hipSuccess) return e;       // (nothing was enqueued when the first half does not apply)
    q.B = p.B - (int)h;
    const float* depth2 = depth + h * p.dc * n;
    const float* value2 = value ? value + h * p.vc * n : nullptr;
    const uint8_t* valid2 = valid ? valid + h * p.valid_c * n : nullptr;
    e = run_window_fused(q, frames_host + h, depth2, value2, valid2, out, mask, 1, ws, ws_bytes, s);
    // (a second half this path refuses -- its split is chosen for its own frame count -- folds in
    // through the generic path: the caller must not start over on a half-built map)
    if (e == hipErrorNotSupported)
      e = run_generic_fused(q, frames_host + h, depth2, value2, valid2, out, mask, 1, ws, s);
    return e;
  }
  if (e != hipSuccess) return e;
  return window_pass(p, st, reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + st.geom_bytes),
                     depth, value, valid, nullptr, nullptr,
                     p.vc ? p.vc : p.dc, p.fill, p.reduction == DM_REDUCE_MAX ? kMax : kMin, slab_bytes, s, out,
                     mask, accumulate);
}

}  // namespace dm

namespace dm {
// which path the calling thread's last projection took, for dm_debug_last_split / _last_path:
// 0 generic (global atomics), 1 LDS windows with host geometry, 2 strip path
void note_split(int pc, int pr, int pd, int path) {
  g_last_split[0] = pc; g_last_split[1] = pr; g_last_split[2] = pd;
  g_last_path = path;
}
}  // namespace dm

extern "C" __attribute__((visibility("default"))) int dm_debug_last_path(void) { return dm::g_last_path; }

extern "C" __attribute__((visibility("default"))) void dm_debug_last_split(int32_t* out4) {
  for (int i = 0; i < 4; ++i) out4[i] = dm::g_last_split[i];
  dm::g_last_split[3] = 0;
}

// Host only (no GPU needed): the split of the image and every (frame, part) window, exactly
// as stage_windows derives them.
extern "C" __attribute__((visibility("default"))) int dm_debug_windows(
    const dm_params* p, const dm_frame* frames, int min_parts, int pd, int32_t* out_parts,
    int32_t* out_windows, size_t window_capacity) {
  using namespace dm;
  if (!p || !frames || !out_parts || p->B < 1 || pd < 1 || pd > 8) return -1;
  const Parts parts = choose_parts(*p, min_parts, pd);
  const int image_parts = parts.pc * parts.pr, nparts = image_parts * pd;
  out_parts[0] = parts.pc; out_parts[1] = parts.pr; out_parts[2] = parts.pd;
  out_parts[3] = parts.wp; out_parts[4] = parts.hp;
  if (!out_windows) return nparts;
  if (window_capacity < (size_t)p->B * nparts) return -1;
  const bool bounded = frustum_bounded(*p);
  std::vector<PartSlopes> slopes(image_parts);
  for (int pr = 0; pr < parts.pr; ++pr)
    for (int pc = 0; pc < parts.pc; ++pc) {
      const int q0 = pc * parts.wp, r0 = pr * parts.hp;
      const int q1 = q0 + parts.wp < p->W ? q0 + parts.wp : p->W;
      const int r1 = r0 + parts.hp < p->H ? r0 + parts.hp : p->H;
      slopes[pr * parts.pc + pc] = part_slopes(*p, q0, q1, r0, r1);
    }
  for (int b = 0; b < p->B; ++b) {
    const FrameAffine fa = frame_affine(*p, frames[b]);
    for (int k = 0; k < pd; ++k) {
      float dlo = p->dmin, dhi = p->dmax;
      if (pd > 1) band_bounds(p->dmin, p->dmax, pd, k, dlo, dhi, parts.geo);
      for (int ip = 0; ip < image_parts; ++ip) {
        const Window w = part_window(*p, fa, slopes[ip], bounded, dlo, dhi);
        int32_t* o = out_windows + ((size_t)b * nparts + k * image_parts + ip) * 4;
        o[0] = w.x0; o[1] = w.z0; o[2] = w.w; o[3] = w.h;
      }
    }
  }
  return nparts;
}

extern "C" __attribute__((visibility("default"))) size_t dm_debug_slab_budget(size_t bytes) {
  const size_t old = dm::g_slab_budget;
  dm::g_slab_budget = bytes;
  return old;
}

extern "C" __attribute__((visibility("default"))) int dm_debug_force_bands(int on) {
  const int old = dm::g_force_bands;
  dm::g_force_bands = on != 0;
  return old;
}

#ifdef DM_STAMPS
extern "C" __attribute__((visibility("default"))) void dm_debug_stamp_buffer(long long* dev) {
  dm::g_stamp_buffer = dev;
}
#endif
