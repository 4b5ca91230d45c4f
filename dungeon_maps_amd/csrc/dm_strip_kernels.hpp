// Device side of the strip path (dm_strip.hip launches it): k_strip_prepare, k_strip_scatter,
// k_strip_combine.  Same pixel arithmetic and LDS-window scatter as k_window_scatter
// (dm_window_kernels.hpp), but
//   * the geometry (windows, cone edges, per-row covers: dm_strip_geometry.hpp) is derived ON
//     THE DEVICE from the frame records, by k_strip_prepare -- nothing of a call is computed
//     on the host, so the launch sequence depends only on pointers and call-wide constants
//     (graph capturable), and with prepared frames it is derived once per set of poses;
//   * every float4 group of a strip's window that no other strip can reach is written
//     straight from LDS to the map; only groups two or more strips can reach go through a
//     slab, and k_strip_combine visits exactly those (a list k_strip_prepare leaves behind);
//   * the rest of the map -- everything outside the hull of the covers on each row -- gets
//     the fill value from the scatter kernel's fill duty.
#pragma once

#include "dm_strip_geometry.hpp"
#include "dm_window_kernels.hpp"

namespace dm {
namespace {

static_assert(sizeof(strip::FrameGeom) == 336, "FrameGeom layout (tests/test_hip_strip.py reads it)");

// LDS of k_strip_scatter, in floats: [window region: slab_stride | 64 scratch cells |
// this strip's row entries: 2 * max_rows | the rows' reach spans: 2 * max_rows | ray slopes of
// the image rows: H].
__host__ __device__ inline size_t strip_lds_bytes(int slab_cells, int max_rows, int H) {
  return ((size_t)slab_cells + 64 + 4 * (size_t)max_rows + (size_t)H + 4) * 4;
}

// L1 geometry of one frame by ONE wave: lane s < kMaxStrips derives strip s (the same calls as
// strip::frame_geometry, the host's serial version), the union window is reduced over the lanes.
// y0..ho: the frame record's yaw entries, translation and offsets.
__device__ inline void strip_geometry_wave(const strip::Cfg& c, float y0, float y2, float y6, float y8,
                                           float tx, float tz, float wo, float ho, int lane,
                                           strip::FrameGeom* g) {
  using namespace strip;
  const int s = lane < kMaxStrips ? lane : kMaxStrips - 1;
  float cx[8], cz[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { cx[k] = c.cxl[s][k]; cz[k] = c.czl[s][k]; }
  const float tmin = c.tmin[s], tmax = c.tmax[s];
  const bool live = (lane < c.P) & (c.live[s] != 0);
  const Pose p = pose_of(c, y0, y2, y6, y8, tx, tz, wo, ho);
  Win16 w;
  Line L, R;
  bool in = false;
  strip_geometry(c, p, cx, cz, tmin, tmax, live, w, L, R, in);
  if (lane < kMaxStrips) { g->win[lane] = w; g->L[lane] = L; g->R[lane] = R; }
  const unsigned long long in_mask = __builtin_amdgcn_ballot_w64(in & (lane < kMaxStrips));
  const bool some = (lane < kMaxStrips) & (w.w > 0);
  int ux0 = some ? w.x0 : 32767, ux1 = some ? w.x0 + w.w : 0;
  int uz0 = some ? w.z0 : 32767, uz1 = some ? w.z0 + w.h : 0;
#pragma unroll
  for (int m = 1; m < kMaxStrips; m <<= 1) {
    ux0 = min(ux0, __shfl_xor(ux0, m, 64)); ux1 = max(ux1, __shfl_xor(ux1, m, 64));
    uz0 = min(uz0, __shfl_xor(uz0, m, 64)); uz1 = max(uz1, __shfl_xor(uz1, m, 64));
  }
  if (lane == 0) {
    g->U = ux1 > ux0 ? Win16{(short)ux0, (short)uz0, (short)(ux1 - ux0), (short)(uz1 - uz0)}
                     : Win16{0, 0, 0, 0};
    g->ok = p.ok;
    g->inside = (int)(in_mask & 0xffu);
  }
}

// What k_strip_prepare leaves in device memory for a batch ("frame tables"): read-only for
// k_strip_scatter, k_strip_combine and the batch fuse.
struct FrameTables {
  Win16* wins;                // (B, kMaxStrips)
  Win16* unions;              // (B)   union windows, x in whole kSpanAlign groups
  int* flags;                 // (B)   FrameGeom::inside
  int* counts;                // (B)   entries of the frame's shared-group list (may exceed list_cap: clamp)
  strip::RowEntry* rows;      // (B, max_rows, P)   per row and strip: cover, owned
  uint2* reach;               // (B, max_rows)      per row: {lo, width} of the hull of the covers
  uint32_t* list;             // (B, list_cap)      the groups of the reach spans nobody owns
};

// An entry of a frame's shared-group list: row of the union window (12 bits), float4 group of the
// union window's row (12 bits), bit s of the top byte: strip s's cover holds the group.
__host__ __device__ inline uint32_t pack_shared(int row, int group, uint32_t hits) {
  return (uint32_t)row | ((uint32_t)group << 12) | (hits << 24);
}
constexpr int kListMaxRows = 4096, kListMaxGroups = 4096;

struct StripPrepArgs {
  const strip::Cfg* cfg;      // device copy (in front of the frame records)
  const float* frames;        // (B, 32) dm_frame records in device memory
  int slab_stride, max_rows, mw, list_cap;
  FrameTables t;
  int* status;                // set non-zero when a frame's geometry does not fit the launch plan
};

constexpr int kPrepThreads = 1024;
constexpr int kPrepLanes = 16;                 // lanes per row of the union window (>= kMaxStrips)
static_assert(kPrepLanes >= strip::kMaxStrips, "one lane per strip");
static_assert(4096 >= kListMaxGroups, "a row's groups fit the LDS list");
constexpr int kPrepSlices = 4;                 // workgroups per frame, each a band of the union window's rows
constexpr int kPrepListLds = 4096;             // list entries a workgroup collects in LDS before it appends them (>= the groups of a row)

// kPrepSlices workgroups per frame: the frame's geometry (wave 0 of each) and, for the
// workgroup's band of rows of the union window, the row table -- cover and owned span of every
// strip (thread = (row, strip), the strips of a row in neighbouring lanes, which exchange their
// covers by shuffles), the hull of the covers per row ("reach": outside it the map holds the
// fill value) and the list of the float4 groups inside the hulls that no strip owns (several
// strips reach them, or none: what is left to combine after the strips have written their own
// groups; collected in LDS, appended with one atomic per workgroup to the frame's list, whose
// counter the host zeroed with the staged copy).  Runs once per set of poses
// (dm_frames_prepare_f32) or in front of the projection kernels (dm_orth_project_f32).
__global__ void __launch_bounds__(kPrepThreads)
k_strip_prepare(StripPrepArgs a) {
  __shared__ strip::FrameGeom geom;
  __shared__ int listed, list_base;
  __shared__ uint32_t found[kPrepListLds];
  const int b = blockIdx.x, slice = blockIdx.y;
  const float* f = a.frames + (size_t)b * 32;
  if (threadIdx.x == 0) listed = 0;
  if (threadIdx.x < 64)
    strip_geometry_wave(*a.cfg, f[10], f[12], f[16], f[18], f[19], f[20], f[21], f[22], (int)threadIdx.x, &geom);
  __syncthreads();
  const int nparts = a.cfg->P;
  Window U = widen(geom.U);
  {   // the union window in whole span groups (covers reach that far)
    const int ux1 = min((U.x0 + U.w + strip::kSpanAlign - 1) & ~(strip::kSpanAlign - 1), a.mw);
    U.x0 &= ~(strip::kSpanAlign - 1);
    U.w = U.w > 0 ? ux1 - U.x0 : 0;
  }
  // a frame that does not fit what the host sized the launches for: flag it and project nothing
  // (cannot happen when the host derived the sizes from these very frames)
  bool fits = geom.ok != 0 && U.h <= a.max_rows && U.h <= kListMaxRows && U.w <= 4 * kListMaxGroups;
  for (int s = 0; s < strip::kMaxStrips; ++s) fits = fits && (int)geom.win[s].w * geom.win[s].h <= a.slab_stride;
  if (!fits) {
    if (threadIdx.x == 0 && slice == 0 && (U.w > 0 || !geom.ok)) atomicOr(a.status, 1);
    U = Window{0, 0, 0, 0};
  }
  if (slice == 0) {
    if (threadIdx.x < strip::kMaxStrips)
      a.t.wins[(size_t)b * strip::kMaxStrips + threadIdx.x] = fits ? geom.win[threadIdx.x] : Win16{0, 0, 0, 0};
    if (threadIdx.x == 0) { a.t.unions[b] = narrow16(U); a.t.flags[b] = fits ? geom.inside : 0; }
  }
  // appends the entries collected in LDS to the frame's list (all threads call it)
  auto append = [&]() {
    __syncthreads();
    const int n = listed < kPrepListLds ? listed : kPrepListLds;
    if (threadIdx.x == 0) list_base = n > 0 ? atomicAdd(a.t.counts + b, n) : 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kPrepThreads)
      if (list_base + i < a.list_cap) a.t.list[(size_t)b * a.list_cap + list_base + i] = found[i];
    if (threadIdx.x == 0 && list_base + n > a.list_cap) atomicOr(a.status, 2);   // (cannot happen: the list holds every group of U)
    __syncthreads();
    if (threadIdx.x == 0) listed = 0;
    __syncthreads();
  };
  // kPrepLanes lanes per row: the first P of them derive the strips' covers (exchanged by
  // shuffles: owned spans, the hull), then all of them walk the hull's groups
  const int sub = (int)threadIdx.x & (kPrepLanes - 1);
  const int lane0 = (int)threadIdx.x & 63 & ~(kPrepLanes - 1);
  // rows per pass: as many as the block has lanes for, and no more than fit the LDS list whatever they hold
  const int groups_per_row = max(U.w >> 2, 1);
  const int per_pass = min(kPrepThreads / kPrepLanes, max(kPrepListLds / groups_per_row, 1));
  const int band = (U.h + kPrepSlices - 1) / kPrepSlices;
  const int r_end = min(U.h, (slice + 1) * band);
  for (int r0 = slice * band; r0 < r_end; r0 += per_pass) {      // (uniform trip count: append() has barriers)
    const int r = r0 + (int)threadIdx.x / kPrepLanes;
    if ((int)threadIdx.x / kPrepLanes < per_pass && r < r_end) {
      const int ps = sub < nparts ? sub : 0;
      uint32_t cover = strip::row_cover(geom.win[ps], geom.L[ps], geom.R[ps], U.z0 + r, a.mw);
      cover = sub < nparts ? cover : 0u;
      uint32_t cov[strip::kMaxStrips], own[strip::kMaxStrips];
#pragma unroll
      for (int q = 0; q < strip::kMaxStrips; ++q) cov[q] = (uint32_t)__shfl((int)cover, lane0 + q, 64);
      int rlo = 32767, rhi = 0;
#pragma unroll
      for (int q = 0; q < strip::kMaxStrips; ++q) {
        rlo = min(rlo, cov[q] ? (int)(cov[q] & 0xffffu) : 32767); rhi = max(rhi, (int)(cov[q] >> 16));
      }
      // the owned span of strip `sub`: its cover cut by the others' in the order sub ^ 1, sub ^ 2, ...
      // (strip::row_owned, the host's version)
      const int P2 = nparts <= 4 ? 4 : 8;
      int lo = (int)(cover & 0xffffu), hi = (int)(cover >> 16);
#pragma unroll
      for (int m = 1; m < strip::kMaxStrips; ++m) {
        const uint32_t other = (uint32_t)__shfl((int)cover, lane0 + ((sub ^ m) & (strip::kMaxStrips - 1)), 64);
        if (m < P2 && (sub ^ m) < nparts) strip::cut_span(lo, hi, other);
      }
      const uint32_t owned = (sub < nparts && hi > lo) ? (uint32_t)lo | ((uint32_t)hi << 16) : 0u;
      if (sub < nparts) a.t.rows[((size_t)b * a.max_rows + r) * nparts + sub] = strip::RowEntry{cover, owned};
      if (sub == 0) a.t.reach[(size_t)b * a.max_rows + r] = rhi > rlo ? make_uint2((unsigned)rlo, (unsigned)(rhi - rlo)) : make_uint2(0u, 0u);
#pragma unroll
      for (int q = 0; q < strip::kMaxStrips; ++q) own[q] = (uint32_t)__shfl((int)owned, lane0 + q, 64);
      // the groups of [rlo, rhi) nobody owns: the row's lanes take kPrepLanes neighbouring groups at
      // a time and append theirs in lane order -- runs of up to kPrepLanes entries whose groups are
      // neighbours in the slabs, so that the combine kernel's threads read them coalesced
      for (int x0 = rlo; x0 < rhi; x0 += 4 * kPrepLanes) {       // (the same trips in all lanes of a row)
        const int x = x0 + 4 * sub;
        uint32_t hits = 0;
        bool mine = false;
#pragma unroll
        for (int q = 0; q < strip::kMaxStrips; ++q) {
          hits |= strip::in_span(cov[q], x) ? 1u << q : 0u;
          mine = mine | strip::in_span(own[q], x);
        }
        const bool want = x < rhi && !mine;
        const unsigned group_bits = (unsigned)((__builtin_amdgcn_ballot_w64(want) >> lane0) & ((1ull << kPrepLanes) - 1));
        const int count = __builtin_popcount(group_bits);
        int base = 0;
        if (sub == 0 && count > 0) base = atomicAdd(&listed, count);
        base = __shfl(base, lane0, 64);
        if (want) {
          const int at = base + __builtin_popcount(group_bits & ((1u << sub) - 1u));
          if (at < kPrepListLds) found[at] = pack_shared(r, (x - U.x0) >> 2, hits);
          else atomicOr(a.status, 2);         // (a pass of rows with more than kPrepListLds shared groups)
        }
      }
    }
    append();
  }
}

struct StripArgs {
  int W, H;
  int clip, flip_h;
  float cx, cy, fx, fy, res;
  float fx_inv, fy_inv, res_inv;
  float dmin, dmax, hmax;
  float Hm1, mhm1;
  int wp, P;
  int fill_parts;             // workgroups per (frame, channel) that share the fill duty: the P strips' plus
                              // fill-only ones (small batches: more workgroups than strips to fill the chip)
  int dc, valid_c;
  int oc, ch0, oc_total;      // channels of this launch's group / first channel / channels of out
  int slab_stride;            // cells per slab = cells of the LDS window region
  int max_rows;               // rows of a frame's row table (>= the union window's height)
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
  const Win16* g_wins;        // frame tables (k_strip_prepare)
  const Win16* g_unions;
  const int* g_flags;
  const strip::RowEntry* g_rows;
  const uint2* g_reach;
  uint16_t* list;             // (B, P, H, wp) cells of the pixels inside their strips' windows (index / value pass)
#ifdef DM_STAMPS
  long long* stamps;
#endif
};

// RED: kMin / kMax.  Always the fast geometry (axis-aligned rotations, exact FMA division),
// 16-byte depth loads.  HAS_VALUE / HAS_VALID / LEAN as in k_window_scatter.
//
// The pixel loop is bound by VALU issue (tools/strip_stamps.py with the DM_X_* switches: 21 us
// of arithmetic against 15 us of memory traffic at cfg2), so everything a pixel does not need
// is kept out of it: the geometry comes from k_strip_prepare, the ray slope of an image row from
// a table in LDS, the fill duty is wave-level stores with scalar addressing, and a strip whose
// window the map's borders did not clip (FrameGeom::inside) skips the window test -- every
// pixel with a depth in range then lands inside the window by construction.
//
// MODE (value maps of many channels: one index computation per pixel, maps.py:314-350):
//   kProject     the whole projection in one kernel (heights; value maps of few channels).
//   kIndexOut    the index pass: no window, no fill, no flush -- every pixel's cell inside its
//                strip's window (16 bits, 0xffff: rejected) goes to a list in the workspace.
//   kFromList    the value pass, one workgroup per (strip, channel, frame): cells from the list,
//                values from the channel's image -- 8 + 16 bytes per thread and row, four
//                instructions per pixel instead of twenty, four rows in flight.
enum { kProject = 0, kIndexOut = 1, kFromList = 2 };
template <int RED, bool HAS_VALID, bool HAS_VALUE, bool LEAN, int MODE = kProject>
__global__ void __launch_bounds__(kScatterThreads)
k_strip_scatter(StripArgs a) {
  constexpr int VEC = 4;
  static_assert(MODE != kFromList || HAS_VALUE, "the value pass scatters values");
  static_assert(MODE != kIndexOut || !HAS_VALUE, "the index pass reads no values");
  // rows of a thread in flight per pipeline stage: value maps carry a second float4 per row and
  // spill at four (the kernel is capped at 128 VGPRs by its 1024 threads)
  constexpr int kRowsInFlight = (HAS_VALUE && MODE != kFromList) ? 2 : dm::kRowsInFlight;
  // Heights without a height test: the camera height is added when the window is flushed, not
  // per pixel.  x -> RN(x + c) is monotone, so max_i RN(h_i + c) = RN(max_i h_i + c) (min
  // alike): the window reduces the raw heights from the reduction's identity, and the flush
  // turns a cell into combine(fill, cell + cam_h) -- which is the fill value where nothing landed.
  constexpr bool kDeferCamH = !HAS_VALUE && LEAN && MODE == kProject;
  // fill steps (wave-level, 1 KB each) per pipeline half-iteration: three or four make every wait for
  // the depth loads wait for older stores too (+2 us per extra step at cfg2), none or one moves
  // the stores behind the loop for the same total
  constexpr int kStripFillPerHalf = 2;
  extern __shared__ float lds[];
  // (the value pass runs channel-major: the workgroups of one (frame, strip) -- which read the same
  // part of the pixel list -- are dispatched together)
  const int part = MODE == kFromList ? blockIdx.y : blockIdx.x;   // column strip
  const int chl = MODE == kFromList ? blockIdx.x : blockIdx.y;    // channel within this launch's group
  const int bl = blockIdx.z, b = a.b0 + bl;
  const int ch = a.ch0 + chl;
  const int dch = a.dc == 1 ? 0 : ch;
  const int nparts = a.P;
  // a fill-only workgroup (part >= P) has no pixels and no window: its strip lies past the image's
  // right edge (nx = 0 below), its window is empty, the tables are read as strip 0's and not used
  const bool fill_only = part >= nparts;
  const int tpart = fill_only ? 0 : part;
  // the strip's pixel rectangle and the first depth rows: kernel arguments only
  const int q0 = part * a.wp;
  int q1 = q0 + a.wp; if (q1 > a.W) q1 = a.W;
  const int r0 = 0, r1 = a.H;
  const int nx = q1 > q0 ? (q1 - q0 + VEC - 1) / VEC : 0;      // (0: a strip past the image's right edge)
  const int ntx = nx < 1 ? 1 : (nx < kScatterThreads ? nx : kScatterThreads);
  const int rows_per_iter = kScatterThreads / ntx;
  const int gx = threadIdx.x % ntx, gy = threadIdx.x / ntx;
  const size_t N = (size_t)a.H * a.W;
  // the images of this (frame, channel) as raw buffer resources: a scalar base and 32-bit offsets
  // instead of 64-bit address arithmetic per row (H * W < 2^28: dm_strip.hip make_plan)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t rs_depth = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.depth) + ((size_t)b * a.dc + dch) * N, 0, (unsigned)N * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_valid = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(HAS_VALID ? a.valid + ((size_t)b * a.valid_c + (a.valid_c == 1 ? 0 : dch)) * N : a.valid),
      0, HAS_VALID ? (unsigned)N : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_value = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(HAS_VALUE ? a.value + ((size_t)b * a.oc_total + ch) * N : a.value), 0,
      HAS_VALUE ? (unsigned)N * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_list = __builtin_amdgcn_make_buffer_rsrc(
      a.list + (MODE != kProject ? ((size_t)b * nparts + tpart) * (size_t)a.H * a.wp : 0), 0,
      MODE != kProject ? (unsigned)a.H * (unsigned)a.wp * 2u : 0u, 0x00020000);
  const float qnan = __builtin_nanf("");
  float za[kRowsInFlight][VEC], zb_[kRowsInFlight][VEC];
  float va[HAS_VALUE ? kRowsInFlight : 1][VEC], vb_[HAS_VALUE ? kRowsInFlight : 1][VEC];
  float aya[kRowsInFlight], ayb[kRowsInFlight];     // the rows' ray slopes (from the LDS table)
  auto load_rows_at = [&](float (&z)[kRowsInFlight][VEC],
                          float (&sv)[HAS_VALUE ? kRowsInFlight : 1][VEC], int q, int r) {
#pragma unroll
    for (int u = 0; u < kRowsInFlight; ++u) {
      int rr = r + u * rows_per_iter;
      rr = rr < r1 ? rr : r1 - 1;              // tail rows repeat the last row (max / min: idempotent)
      const int at = __mul24(rr, a.W) + q;     // (pixel index inside the image; 24-bit multiply: full rate)
      if (MODE == kFromList) {                 // four 16-bit cells instead of four depths
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(rs_list, (__mul24(rr, a.wp) + (q - q0)) << 1, 0, 0);
        z[u][0] = __uint_as_float(c.x); z[u][1] = __uint_as_float(c.y);
      } else {
        const f32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_depth, at << 2, 0, 0);
        z[u][0] = t.x; z[u][1] = t.y; z[u][2] = t.z; z[u][3] = t.w;
      }
      if (HAS_VALID && MODE != kFromList) {                         // (four bools at once: q is a multiple of 4)
        const unsigned ok4 = __builtin_amdgcn_raw_buffer_load_b32(rs_valid, at, 0, 0);
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          z[u][k] = ((ok4 >> (8 * k)) & 0xffu) ? z[u][k] : qnan;
      }
      if (HAS_VALUE) {
        const f32x4 s = __builtin_amdgcn_raw_buffer_load_b128(rs_value, at << 2, 0, 0);
        sv[u][0] = s.x; sv[u][1] = s.y; sv[u][2] = s.z; sv[u][3] = s.w;
      }
    }
  };
  // the first rows: requested from kernel arguments alone, before anything else
  bool first_rows_loaded = false;
  if (gx < nx && q1 > q0) {
    load_rows_at(za, va, q0 + gx * VEC, r0 + gy);
    first_rows_loaded = true;
  }

  // the frame's record and tables: one batch of scalar loads, pinned
  const float* tf = a.frames + (size_t)b * 32;
  float p4 = tf[4], p5 = tf[5], p7 = tf[7], p8 = tf[8], cam_h = tf[9];
  float fy0 = tf[10], fy2 = tf[12], fy6 = tf[16], fy8 = tf[18], ftx = tf[19], ftz = tf[20];
  float wo = tf[21], ho = tf[22];
  // this strip's row entries (cover, owned span) and the rows' reach spans: requested right behind
  // the first depth rows, for as many rows as the table holds (the union window's height is not
  // known yet, and waiting for it would put two round trips in a row into the kernel's head)
  strip::RowEntry row_e = {0u, 0u};
  uint2 reach_e = make_uint2(0u, 0u);
  if ((int)threadIdx.x < a.max_rows) {
    row_e = a.g_rows[((size_t)b * a.max_rows + threadIdx.x) * nparts + tpart];
    reach_e = a.g_reach[(size_t)b * a.max_rows + threadIdx.x];
  }
  const int2 w_raw = *reinterpret_cast<const int2*>(a.g_wins + (size_t)b * strip::kMaxStrips + tpart);
  const int2 u_raw = *reinterpret_cast<const int2*>(a.g_unions + b);
  int flags = a.g_flags[b];
#ifdef DM_STAMPS
  long long stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DM_STAMP(0);
  const int table_off = a.slab_stride + 64;
  strip::RowEntry* rows = reinterpret_cast<strip::RowEntry*>(lds + table_off);   // this strip's entry of every row of U
  uint2* reach = reinterpret_cast<uint2*>(lds + table_off + 2 * a.max_rows);     // {lo, width} of every row of U
  float* aytab = lds + table_off + 4 * a.max_rows;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  // The whole window region of LDS gets the fill value, the ray-slope table its H entries
  // (maps.py:670-678; border rows poisoned), under the first depth rows in flight.
  const float lds_init = kDeferCamH ? (RED == kMax ? -INFINITY : INFINITY) : a.fill;
  if (MODE != kIndexOut)
    for (int i = threadIdx.x * 4; i < a.slab_stride + 64; i += kScatterThreads * 4)
      *reinterpret_cast<float4*>(lds + i) = make_float4(lds_init, lds_init, lds_init, lds_init);
  for (int r = threadIdx.x; r < a.H; r += kScatterThreads) {
    float yr = (float)r;
    yr = a.flip_h ? a.Hm1 - yr : yr;
    float ay = div_markstein(yr - a.cy, a.fy, a.fy_inv);
    if (!LEAN) ay = (r < a.clip || r >= a.H - a.clip) ? qnan : ay;
    aytab[r] = ay;
  }
  DM_STAMP(1);
  // (the scalar loads are waited for only here, under the LDS work above; pinned: one batch)
  asm volatile("" : "+s"(p4), "+s"(p5), "+s"(p7), "+s"(p8), "+s"(cam_h), "+s"(fy0), "+s"(fy2),
                    "+s"(fy6), "+s"(fy8), "+s"(ftx), "+s"(ftz), "+s"(wo), "+s"(ho));
  Window w = {(short)(w_raw.x & 0xffff), (short)(w_raw.x >> 16), (short)(w_raw.y & 0xffff), (short)(w_raw.y >> 16)};
  Window U = {(short)(u_raw.x & 0xffff), (short)(u_raw.x >> 16), (short)(u_raw.y & 0xffff), (short)(u_raw.y >> 16)};
  {
    // wave-uniform values: keep them in SGPRs
    w.x0 = __builtin_amdgcn_readfirstlane(w.x0); w.z0 = __builtin_amdgcn_readfirstlane(w.z0);
    w.w = __builtin_amdgcn_readfirstlane(fill_only ? 0 : w.w); w.h = __builtin_amdgcn_readfirstlane(w.h);   // (fill-only: no window)
    U.x0 = __builtin_amdgcn_readfirstlane(U.x0); U.z0 = __builtin_amdgcn_readfirstlane(U.z0);
    U.w = __builtin_amdgcn_readfirstlane(U.w); U.h = __builtin_amdgcn_readfirstlane(U.h);
    flags = __builtin_amdgcn_readfirstlane(flags);
  }
  const int area = w.w * w.h;
  // the row tables -> LDS (rows past the union window's height hold whatever the table does: never used)
  if ((int)threadIdx.x < a.max_rows) { rows[threadIdx.x] = row_e; reach[threadIdx.x] = reach_e; }
  for (int r = threadIdx.x + kScatterThreads; r < U.h; r += kScatterThreads) {
    rows[r] = a.g_rows[((size_t)b * a.max_rows + r) * nparts + tpart];
    reach[r] = a.g_reach[(size_t)b * a.max_rows + r];
  }
  lds_barrier();
  DM_STAMP(2);
  DM_STAMP(3);
  // Fill duty: map rows part, part + F, ... (F = fill_parts) of (b, ch) outside the rows' reach spans (the hull of
  // the strips' covers: inside it the flush below and the combine step write).  Wave-level: wave v takes the rows part + (v + 16 j) F,
  // one step stores 256 cells of a row (float4 per lane) and their mask bytes with SCALAR
  // addressing -- the map of this (frame, channel) as a raw buffer resource, the row and chunk
  // in the scalar offset, the lane's fixed 16 / 4 bytes in the vector offset.  A lane that has
  // nothing to write (inside the reach, past the row's end, past the wave's rows) gets a vector offset
  // past the end of the buffer and is dropped by the hardware's range check: no branch in the
  // loop, a handful of VALU instructions per KB.
  const size_t map_base = ((size_t)b * a.oc_total + ch) * (size_t)a.mh * a.mw;
  const int fparts = a.fill_parts;
  const int fill_rows = (a.mh - part + fparts - 1) / fparts;
  const int chunks = (a.mw + 255) >> 8;
#ifdef DM_X_NOFILL2
  const int fill_steps = 0;
#else
  const int fill_steps = a.out != nullptr && wave < fill_rows ? ((fill_rows - wave + 15) >> 4) * chunks : 0;
#endif
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const unsigned map_cells = (unsigned)a.mh * (unsigned)a.mw;
  const __amdgpu_buffer_rsrc_t rs_out =
      __builtin_amdgcn_make_buffer_rsrc(a.out + map_base, 0, a.out ? map_cells * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mask =
      __builtin_amdgcn_make_buffer_rsrc(a.mask + map_base, 0, a.out ? map_cells : 0u, 0x00020000);
  const unsigned fill_bits = __float_as_uint(a.fill);
  const int lane4 = lane << 2;
  int fs = 0, f_row = part + wave * fparts, f_chunk = 0;       // (wave-uniform)
  // the reach of a map row ({0, 0} outside U's rows), read from LDS TWO steps ahead: the steps
  // come in pairs, and a value read one step ahead would make the second step of a pair wait
  // for every LDS operation in flight (one counter), the pixel loop's atomics among them
  auto reach_of = [&](int row) {
    const int i = row - U.z0;
    const bool in_rows = (unsigned)i < (unsigned)U.h;                               // (scalar)
    const uint2 rr = reach[in_rows ? i : 0];
    return make_uint2(rr.x, in_rows ? rr.y : 0u);
  };
  auto advance = [&](int& row, int& chunk) {
    const bool next_row = chunk + 1 == chunks;
    chunk = next_row ? 0 : chunk + 1;
    row += next_row ? 16 * fparts : 0;
  };
  int f_row2 = f_row, f_chunk2 = f_chunk;      // where the fill duty is two steps from now
  uint2 f_reach = reach_of(f_row2);
  advance(f_row2, f_chunk2);
  uint2 f_reach1 = reach_of(f_row2);
  advance(f_row2, f_chunk2);
  auto fill_step = [&]() {
    const int x = (f_chunk << 8) + lane4;
    const bool live = fs < fill_steps;                                              // (scalar)
    const bool skip = !live | (x >= a.mw) | ((unsigned)(x - (int)f_reach.x) < f_reach.y);
    // (the scalar offset must be the same in every lane, skipping or not: a lane-dependent one
    // costs a waterfall loop per store)
    const int cell0 = __builtin_amdgcn_readfirstlane(live ? f_row * a.mw + (f_chunk << 8) : 0);
    __builtin_amdgcn_raw_buffer_store_b128((u32x4){fill_bits, fill_bits, fill_bits, fill_bits}, rs_out,
                                           skip ? 0x7ffffff0 : lane4 << 2, cell0 << 2, 0);
    __builtin_amdgcn_raw_buffer_store_b32(0u, rs_mask, skip ? 0x7ffffff0 : lane4, cell0, 0);
    ++fs;
    advance(f_row, f_chunk);
    f_reach = f_reach1;
    f_reach1 = reach_of(f_row2);
    advance(f_row2, f_chunk2);
  };
  // (a local map's records carry a neutral yaw and no translation: dm_strip.hip stage_frames)
  const float y0 = fy0, y2r = fy2, y6 = fy6, y8 = fy8, tx = ftx, tz = ftz;
  const float flip_s = a.flip_h ? -1.0f : 1.0f, flip_c = a.flip_h ? a.mhm1 : 0.0f;
  // LDS addresses of the pixel loop as plain 32-bit byte addresses (an LDS pointer IS that), so
  // that the window's origin and the base of `lds` fold into one scalar:
  // address = (z * w + x) * 4 + origin
  typedef __attribute__((address_space(3))) float lds_float;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_float*)lds;
  const unsigned dummy = lds_base + (((unsigned)a.slab_stride + (unsigned)lane) << 2);   // 64 scratch cells
  const int origin = (int)lds_base - 4 * (w.z0 * w.w + w.x0);
  auto lds_at = [](unsigned addr) { return (lds_float*)(uintptr_t)addr; };

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
      // (the ray slopes are read from LDS together with the depth loads, a whole pipeline stage
      // ahead: read where they are used they would make every row wait for the LDS atomics of
      // the row before -- one counter)
      auto load_ay = [&](float (&ay)[kRowsInFlight], int r) {
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
          int rr = r + u * rows_per_iter;
          rr = rr < r1 ? rr : r1 - 1;
          ay[u] = aytab[rr];                                                 // maps.py:670-678
        }
      };
      auto project_rows = [&](auto tested, const float (&z)[kRowsInFlight][VEC],
                              const float (&sv)[HAS_VALUE ? kRowsInFlight : 1][VEC],
                              const float (&ayr)[kRowsInFlight], int r) {
        constexpr bool kTest = decltype(tested)::value;
        if (MODE == kFromList) {
#pragma unroll
          for (int u = 0; u < kRowsInFlight; ++u) {
            const unsigned c01 = __float_as_uint(z[u][0]), c23 = __float_as_uint(z[u][1]);
            const unsigned cell[VEC] = {c01 & 0xffffu, c01 >> 16, c23 & 0xffffu, c23 >> 16};
            unsigned li[VEC];
            float hv[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
              const float sval = sv[u][k];
              const bool ok = (cell[k] != 0xffffu) & (sval == sval);        // NaN never replaces a number
              li[k] = ok ? lds_base + (cell[k] << 2) : dummy;
              hv[k] = sval;
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
#ifndef DM_X_NOATOMIC
#pragma unroll
            for (int k = 0; k < VEC; ++k) lds_reduce<RED>(lds_at(li[k]), hv[k]);
#else
            if (li[0] + li[1] + li[2] + li[3] == 12345u && hv[0] + hv[1] + hv[2] + hv[3] == 1.5f) lds_reduce<RED>(lds_at(dummy), hv[0]);
#endif
          }
          return;
        }
#ifdef DM_X_NOMATH
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u)
          lds_reduce<RED>(lds_at(dummy), fmaxf(fmaxf(z[u][0], z[u][1]), fmaxf(z[u][2], z[u][3])));
        return;
#endif
        // (LEAN heights: the two depth compares of a pixel go straight into the execution mask of
        // its LDS atomic -- v_cmpx twice, ds_max, mask back -- instead of into a select between
        // the cell and a dummy cell: 1.25 instructions per pixel less.  The LDS operation inside
        // the asm is invisible to the compiler's wait counting, which can only make it wait
        // longer: LDS operations complete in order, and it counts fewer younger ones than there are.)
        constexpr bool kExecMask = LEAN && !HAS_VALUE && MODE == kProject;
        unsigned long long exec_all = 0;
        if (kExecMask) asm volatile("s_mov_b64 %0, exec" : "=s"(exec_all));
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
          const float ay = ayr[u];
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
            f2 h1 = __builtin_elementwise_fma(zz, (f2){p7, p7}, Y * p4);                  // maps.py:790-797
            if (!kDeferCamH) h1 = h1 + cam_h;
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
            const int ix = floor_to_int(xf), iz = floor_to_int(zf);
            // (bitwise &: the short-circuit form compiles to a branch per pixel, and a branch in
            // this loop costs its counted waits)
            bool ok = (zz >= a.dmin) & (zz <= a.dmax);
            if (kTest) ok = ok & ((unsigned)(ix - w.x0) < (unsigned)w.w) & ((unsigned)(iz - w.z0) < (unsigned)w.h);
            if (!LEAN) ok = ok & !__builtin_isunordered(xf, zf) & (h1 <= a.hmax);
            const float sval = HAS_VALUE ? sv[u][k] : h1;
            if (HAS_VALUE) ok = ok & (sval == sval);             // NaN never replaces a number
            // (window coordinates fit 24 bits: |cell coordinates| < 2^23, dm_strip.hip validate_frames)
            unsigned addr = (unsigned)(((__mul24(iz, w.w) + ix) << 2) + origin);
            asm("" : "+v"(addr));
            li[k] = ok ? addr : dummy;
            if (kExecMask && !kTest) li[k] = addr;       // (validity is applied by the execution mask below)
            if (MODE == kIndexOut) li[k] = ok ? (addr - lds_base) >> 2 : 0xffffu;    // the cell inside the window
            hv[k] = sval;
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
          if (MODE == kIndexOut) {
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            int rr = r + u * rows_per_iter;
            rr = rr < r1 ? rr : r1 - 1;
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){li[0] | (li[1] << 16), li[2] | (li[3] << 16)}, rs_list,
                                                  (__mul24(rr, a.wp) + (q - q0)) << 1, 0, 0);
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
      const int niter = (r1 - r0 + step - 1) / step;
      // Two copies of the pipelined loop -- with and without the window test -- chosen by ONE
      // wave-uniform branch: inside, exactly kStripFillPerHalf unconditional fill steps follow each
      // group of loads, so the waits count them and never wait on a store or on the prefetch.
      auto pipeline = [&](auto tested) {
        int r = r0 + gy;
        if (!first_rows_loaded) load_rows(za, va, r);
        first_rows_loaded = false;
        load_ay(aya, r);
        load_rows(zb_, vb_, r + step);
        load_ay(ayb, r + step);
        DM_STAMP(11);
        // younger waves of a SIMD get the higher issue priority in the loop (age arbitration
        // favours the oldest wave otherwise, and the last wave left on a SIMD runs latency
        // bound).  Only now: under static priorities the four waves of a SIMD run the code
        // above one after the other instead of hiding each other's latencies.
#ifdef DM_X_ROTPRIO
        const int cls = wave >> 2;
        // s_setprio takes an immediate: one opaque block of scalar code picks it (a branch the
        // compiler sees would make it drain the loop's counted waits)
        auto set_prio = [](int p) {
          asm volatile("s_cmp_eq_u32 %0, 3\n\ts_cbranch_scc1 .Lp3_%=\n\t"
                       "s_cmp_eq_u32 %0, 2\n\ts_cbranch_scc1 .Lp2_%=\n\t"
                       "s_cmp_eq_u32 %0, 1\n\ts_cbranch_scc1 .Lp1_%=\n\t"
                       "s_setprio 0\n\ts_branch .Lpe_%=\n"
                       ".Lp3_%=:\n\ts_setprio 3\n\ts_branch .Lpe_%=\n"
                       ".Lp2_%=:\n\ts_setprio 2\n\ts_branch .Lpe_%=\n"
                       ".Lp1_%=:\n\ts_setprio 1\n"
                       ".Lpe_%=:\n" :: "s"(p) : "scc");
        };
#elif !defined(DM_X_NOPRIO)
        if (wave >= 12) __builtin_amdgcn_s_setprio(3);
        else if (wave >= 8) __builtin_amdgcn_s_setprio(2);
        else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
        for (int it = 0; it < niter; it += 2) {
#ifdef DM_X_ROTPRIO
          set_prio((cls + it) & 3);
#endif
#pragma unroll
          for (int t = 0; t < kStripFillPerHalf; ++t) fill_step();
          project_rows(tested, za, va, aya, r);
          if (it + 1 < niter) {
            load_rows(za, va, r + 2 * step);
            load_ay(aya, r + 2 * step);
#ifdef DM_X_ROTPRIO
            set_prio((cls + it + 1) & 3);
#endif
#pragma unroll
            for (int t = 0; t < kStripFillPerHalf; ++t) fill_step();
            project_rows(tested, zb_, vb_, ayb, r + step);
            load_rows(zb_, vb_, r + 3 * step);        // (past the end: the last row again, unused)
            load_ay(ayb, r + 3 * step);
          }
          r += 2 * step;
        }
      };
      if ((flags >> part) & 1) pipeline(std::false_type{}); else pipeline(std::true_type{});
    }
  }
  DM_STAMP(4);
#ifdef DM_STAMPS
  if (a.stamps && (threadIdx.x & 63) == 0)     // every wave: when it left the pixel loop
    a.stamps[4096 * 12 + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + wave] = stamp[4];
#endif
  while (fs < fill_steps) fill_step();
  __builtin_amdgcn_s_setprio(0);
  lds_barrier();
  DM_STAMP(5);
  // Flush, 16 lanes per window row: the groups of this strip's cover go straight to the map where
  // the strip owns them (no other strip's cover reaches them), else to the slab: k_strip_combine
  // combines those with the other strips'.
  if (area > 0 && MODE != kIndexOut) {
    const int l16 = (int)threadIdx.x & 15;
    const int unit = b * a.oc + chl;             // (frame, channel): the P workgroups that share a map
    float* slab = a.slabs + ((size_t)unit * nparts + part) * a.slab_stride;
    for (int row = (int)threadIdx.x >> 4; row < w.h; row += kScatterThreads / 16) {
      const int z = w.z0 + row;
      const strip::RowEntry e = rows[z - U.z0];
      const int lo = (int)(e.cover & 0xffffu), hi = (int)(e.cover >> 16);
      const int cell0 = row * w.w - w.x0;
      for (int x = lo + (l16 << 2); x < hi; x += 64) {
        float4 v = *reinterpret_cast<const float4*>(lds + cell0 + x);
        if (kDeferCamH) {
          v.x = combine<RED>(v.x + cam_h, a.fill); v.y = combine<RED>(v.y + cam_h, a.fill);
          v.z = combine<RED>(v.z + cam_h, a.fill); v.w = combine<RED>(v.w + cam_h, a.fill);
        }
        if (strip::in_span(e.owned, x)) {
          const int cell = z * a.mw + x;
          __builtin_amdgcn_raw_buffer_store_b128((f32x4){v.x, v.y, v.z, v.w}, rs_out, cell << 2, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(
              (uint32_t)mask_of(v.x, a.fill) | ((uint32_t)mask_of(v.y, a.fill) << 8) |
              ((uint32_t)mask_of(v.z, a.fill) << 16) | ((uint32_t)mask_of(v.w, a.fill) << 24), rs_mask, cell, 0, 0);
        } else {
          *reinterpret_cast<float4*>(slab + cell0 + x) = v;
        }
      }
    }
  }
  DM_STAMP(6);
  DM_STAMPS_OUT();
}

struct StripCombineArgs {
  int b0, oc, ch0, oc_total, mh, mw;
  int P, slab_stride, list_cap;
  float fill;
  const Win16* g_wins;
  const Win16* g_unions;
  const int* g_counts;
  const uint32_t* g_list;
  const float* slabs;
  float* out;
  uint8_t* mask;
};

constexpr int kCombineThreads = 256;
// A (frame, channel) gets kCombineSlots / ENTRIES blocks whose threads take ENTRIES list entries at a
// time (their slab loads in flight together) and stride over the list.  One entry per thread for
// height maps (a handful of blocks per frame: the kernel is one chain of round trips, 6 us at
// cfg2), four for value maps of many channels (the chip holds 2 K blocks at a time: with one
// entry per thread 40 channels were twenty rounds of that chain, 200 us; with four, 80 us).
constexpr int kCombineSlots = 16;

// The groups of a frame's reach spans that no strip owns (the frame's shared-group list,
// k_strip_prepare): max / min over the slabs of the strips whose covers hold the group -- the fill
// value where none does -- written to the map with its mask bytes.  kCombineEntries list entries per thread at a time;
// what a block needs of the frame (count, union window, the strips' windows) is wave-uniform.
// One list entry per thread (height maps).
template <int RED>
__global__ void __launch_bounds__(kCombineThreads)
k_strip_combine_one(StripCombineArgs a) {
  constexpr int kCombineBlocks = kCombineSlots;
  const int fcl = blockIdx.y;                  // (frame of the launch) * oc + channel of the group
  const int bl = fcl / a.oc, b = a.b0 + bl;
  const int chl = fcl - bl * a.oc;
  const int first = blockIdx.x * kCombineThreads + (int)threadIdx.x;
  const uint32_t* const list = a.g_list + (size_t)b * a.list_cap;
  // (the thread's first entry is requested before the list's length is known: one round trip
  // less in a kernel that is nothing but a chain of them)
  uint32_t entry = first < a.list_cap ? list[first] : 0u;
  const int listed = min(a.g_counts[b], a.list_cap);
  if (blockIdx.x * kCombineThreads >= listed) return;
  const int2 u_raw = *reinterpret_cast<const int2*>(a.g_unions + b);
  const int ux0 = (short)(u_raw.x & 0xffff), uz0 = (short)(u_raw.x >> 16);
  int2 wq[strip::kMaxStrips];
#pragma unroll
  for (int q = 0; q < strip::kMaxStrips; ++q)
    wq[q] = *reinterpret_cast<const int2*>(a.g_wins + (size_t)b * strip::kMaxStrips + (q < a.P ? q : 0));
  const float* const slabs = a.slabs + ((size_t)(b * a.oc + chl) * a.P) * a.slab_stride;
  const size_t fo = ((size_t)b * a.oc_total + a.ch0 + chl) * (size_t)a.mh * a.mw;
  const float ident = RED == kMax ? -INFINITY : INFINITY;
  for (int i = first; i < listed; i += kCombineBlocks * kCombineThreads) {
    if (i != first) entry = list[i];
    const int z = uz0 + (int)(entry & 0xfffu), x = ux0 + (int)(((entry >> 12) & 0xfffu) << 2);
    float4 t[strip::kMaxStrips];
#pragma unroll
    for (int q = 0; q < strip::kMaxStrips; ++q) {
      const bool hit = ((entry >> (24 + q)) & 1u) != 0u;      // (never set for q >= P)
      const int wx0 = (short)(wq[q].x & 0xffff), wz0 = (short)(wq[q].x >> 16), ww = (short)(wq[q].y & 0xffff);
      t[q] = make_float4(ident, ident, ident, ident);
      if (hit) t[q] = *reinterpret_cast<const float4*>(slabs + (size_t)q * a.slab_stride + (z - wz0) * ww + (x - wx0));
    }
    // (the fill value takes part: utils.py:470-477 reduces INTO the filled canvas)
    float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
#pragma unroll
    for (int q = 0; q < strip::kMaxStrips; ++q) {
      acc.x = combine<RED>(acc.x, t[q].x); acc.y = combine<RED>(acc.y, t[q].y);
      acc.z = combine<RED>(acc.z, t[q].z); acc.w = combine<RED>(acc.w, t[q].w);
    }
    const size_t cell = fo + (size_t)z * a.mw + x;
    *reinterpret_cast<float4*>(a.out + cell) = acc;
    *reinterpret_cast<uint32_t*>(a.mask + cell) =
        (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
        ((uint32_t)mask_of(acc.z, a.fill) << 16) | ((uint32_t)mask_of(acc.w, a.fill) << 24);
  }
}

// kCombineEntries list entries per thread at a time (value maps of many channels).
template <int RED, int kCombineEntries>
__global__ void __launch_bounds__(kCombineThreads)
k_strip_combine(StripCombineArgs a) {
  constexpr int kCombineBlocks = kCombineSlots / kCombineEntries;
  const int fcl = blockIdx.y;                  // (frame of the launch) * oc + channel of the group
  const int bl = fcl / a.oc, b = a.b0 + bl;
  const int chl = fcl - bl * a.oc;
  constexpr int kStride = kCombineBlocks * kCombineThreads;
  const int first = blockIdx.x * kCombineThreads + (int)threadIdx.x;
  const uint32_t* const list = a.g_list + (size_t)b * a.list_cap;
  // (the thread's first entries are requested before the list's length is known: one round trip
  // less in a kernel that is nothing but a chain of them)
  uint32_t ent[kCombineEntries];
#pragma unroll
  for (int k = 0; k < kCombineEntries; ++k) ent[k] = first + k * kStride < a.list_cap ? list[first + k * kStride] : 0u;
  const int listed = min(a.g_counts[b], a.list_cap);
  if (blockIdx.x * kCombineThreads >= listed) return;
  const int2 u_raw = *reinterpret_cast<const int2*>(a.g_unions + b);
  const int ux0 = (short)(u_raw.x & 0xffff), uz0 = (short)(u_raw.x >> 16);
  int2 wq[strip::kMaxStrips];
#pragma unroll
  for (int q = 0; q < strip::kMaxStrips; ++q)
    wq[q] = *reinterpret_cast<const int2*>(a.g_wins + (size_t)b * strip::kMaxStrips + (q < a.P ? q : 0));
  const float* const slabs = a.slabs + ((size_t)(b * a.oc + chl) * a.P) * a.slab_stride;
  const size_t fo = ((size_t)b * a.oc_total + a.ch0 + chl) * (size_t)a.mh * a.mw;
  for (int base = first; base < listed; base += kCombineEntries * kStride) {
    if (base != first) {
#pragma unroll
      for (int k = 0; k < kCombineEntries; ++k) ent[k] = base + k * kStride < listed ? list[base + k * kStride] : 0u;
    }
    // where each group lies in the slabs of the (usually two) strips whose covers hold it; the
    // slab loads of all of the thread's entries are in flight together.  Further strips (around
    // the camera's cell) take the slow lane.
    int z[kCombineEntries], x[kCombineEntries];
    uint32_t more[kCombineEntries];
    float4 ta[kCombineEntries], tb[kCombineEntries];
    bool has_a[kCombineEntries], has_b[kCombineEntries];
    auto slab_offset = [&](int k, int q) {
      const int wx0 = (short)(wq[q].x & 0xffff), wz0 = (short)(wq[q].x >> 16), ww = (short)(wq[q].y & 0xffff);
      return q * a.slab_stride + (z[k] - wz0) * ww + (x[k] - wx0);
    };
#pragma unroll
    for (int k = 0; k < kCombineEntries; ++k) {
      const bool live = base + k * kStride < listed;
      const uint32_t hits = live ? ent[k] >> 24 : 0u;          // (never set for strips >= P)
      z[k] = uz0 + (int)(ent[k] & 0xfffu); x[k] = ux0 + (int)(((ent[k] >> 12) & 0xfffu) << 2);
      int off_a = 0, off_b = 0;
      has_a[k] = false; has_b[k] = false; more[k] = 0;
#pragma unroll
      for (int q = 0; q < strip::kMaxStrips; ++q) {
        const bool hit = ((hits >> q) & 1u) != 0u;
        const int o = slab_offset(k, q);
        const bool first_hit = hit & !has_a[k], second_hit = hit & !first_hit & !has_b[k];
        more[k] |= (hit & !first_hit & !second_hit) ? 1u << q : 0u;
        off_a = first_hit ? o : off_a; has_a[k] = has_a[k] | first_hit;
        off_b = second_hit ? o : off_b; has_b[k] = has_b[k] | second_hit;
      }
      ta[k] = *reinterpret_cast<const float4*>(slabs + off_a);     // (offset 0 when there is no hit: discarded)
      tb[k] = *reinterpret_cast<const float4*>(slabs + off_b);
    }
#pragma unroll
    for (int k = 0; k < kCombineEntries; ++k) {
      if (base + k * kStride >= listed) continue;
      // (the fill value takes part: utils.py:470-477 reduces INTO the filled canvas)
      float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
      if (has_a[k]) { acc.x = combine<RED>(acc.x, ta[k].x); acc.y = combine<RED>(acc.y, ta[k].y);
                      acc.z = combine<RED>(acc.z, ta[k].z); acc.w = combine<RED>(acc.w, ta[k].w); }
      if (has_b[k]) { acc.x = combine<RED>(acc.x, tb[k].x); acc.y = combine<RED>(acc.y, tb[k].y);
                      acc.z = combine<RED>(acc.z, tb[k].z); acc.w = combine<RED>(acc.w, tb[k].w); }
      for (uint32_t m = more[k]; m != 0; m &= m - 1) {
        const float4 t = *reinterpret_cast<const float4*>(slabs + slab_offset(k, __builtin_ctz(m)));
        acc.x = combine<RED>(acc.x, t.x); acc.y = combine<RED>(acc.y, t.y);
        acc.z = combine<RED>(acc.z, t.z); acc.w = combine<RED>(acc.w, t.w);
      }
      const size_t cell = fo + (size_t)z[k] * a.mw + x[k];
      *reinterpret_cast<float4*>(a.out + cell) = acc;
      *reinterpret_cast<uint32_t*>(a.mask + cell) =
          (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
          ((uint32_t)mask_of(acc.z, a.fill) << 16) | ((uint32_t)mask_of(acc.w, a.fill) << 24);
    }
  }
}

// Test hook kernel: the geometry of every frame exactly as k_strip_scatter derives it.
__global__ void __launch_bounds__(64)
k_strip_geometry_dump(const strip::Cfg* cfg, const float* frames, strip::FrameGeom* out) {
  __shared__ strip::FrameGeom g;
  const float* f = frames + (size_t)blockIdx.x * 32;
  strip_geometry_wave(*cfg, f[10], f[12], f[16], f[18], f[19], f[20], f[21], f[22], (int)threadIdx.x, &g);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = g;
}

}  // namespace
}  // namespace dm
