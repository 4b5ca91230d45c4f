// Device side of the strip path (dm_strip.hip launches it): k_strip_scatter and k_strip_combine.
// Same pixel arithmetic and LDS-window scatter as k_window_scatter (dm_window_kernels.hpp), but
//   * a launch is STATELESS: the camera state of its (up to kPoseFrames) frames travels in the
//     kernel arguments (48 bytes per frame: yaw entries, translation, offsets, camera height),
//     or is read from a caller's device buffer (prepared frames) -- no table is staged to the
//     device and no kernel runs in front of the scatter kernel;
//   * the geometry (windows, cone edges, per-row covers: dm_strip_geometry.hpp) is derived by
//     every workgroup for its own frame, in the kernel's head, straight into LDS;
//   * every float4 group of a strip's window that no other strip can reach is written
//     straight from LDS to the map; only groups two or more strips can reach go through a
//     slab, and k_strip_combine visits exactly those -- the strips list them while they flush;
//   * the rest of the map -- everything outside the hull of the covers on each row -- gets
//     the fill value from the scatter kernel's fill duty.
#pragma once

#include <stddef.h>

#include "dm_strip_geometry.hpp"
#include "dm_window_kernels.hpp"

namespace dm {
namespace {

static_assert(sizeof(strip::FrameGeom) == 336, "FrameGeom layout (tests/test_hip_strip.py reads it)");

// An entry of a strip's shared-group list: row of the union window (12 bits), float4 group of the
// union window's row (12 bits), bit s of the top byte: strip s's cover holds the group.
__host__ __device__ inline uint32_t pack_shared(int row, int group, uint32_t hits) {
  return (uint32_t)row | ((uint32_t)group << 12) | (hits << 24);
}
constexpr int kListMaxRows = 4096, kListMaxGroups = 4096;

// Camera state of one frame as the kernels read it (dm_frame without what the batch shares:
// the pitch rotation lives in StripArgs).
constexpr int kPoseFloats = 12;
constexpr int kPoseFrames = 64;                // frames per launch: 3 KB of kernel arguments
struct StripPose { float y0, y2, y6, y8, tx, tz, wo, ho, cam_h, pad0, pad1, pad2; };
static_assert(sizeof(StripPose) == kPoseFloats * 4, "pose record");

// L1 geometry of one frame as k_strip_scatter keeps it in LDS (the values of strip::FrameGeom,
// laid out for 16-byte reads).
struct alignas(16) GeomLds {
  Win16 win[strip::kMaxStrips];
  strip::Line L[strip::kMaxStrips], R[strip::kMaxStrips];
  Win16 U;                    // bounding box of the windows
  int ok, inside;             // FrameGeom::ok, FrameGeom::inside
  // list entries of this strip per chunk of 64 rows of the union window (then: before the chunk);
  // [kListMaxRows / 64]: all of them
  int chunk_entries[kListMaxRows / 64 + 1];
  int has_holes;              // some row of this workgroup's fill duty has groups of its hull in no strip's cover
  int pad[2];
};
constexpr int kGeomFloats = sizeof(GeomLds) / 4;
static_assert(sizeof(GeomLds) % 16 == 0, "GeomLds size");

// LDS of k_strip_scatter, in floats: [window region: slab_stride | 64 scratch cells | the
// covers of all strips on every row of the union window: (4 or 8) * max_rows | this strip's owned
// spans: max_rows | the rows' reach spans: 2 * max_rows | where each row's list entries go:
// max_rows | ray slopes of the image rows: H | the frame's geometry and list bookkeeping (GeomLds)].
__host__ __device__ inline int strips_pow2(int P) { return P <= 4 ? 4 : 8; }     // row stride of the cover table
__host__ __device__ inline size_t strip_lds_bytes(int slab_cells, int max_rows, int H, int P) {
  return ((size_t)slab_cells + 64 + (size_t)(strips_pow2(P) + 4) * (size_t)max_rows + (size_t)H + 4) * 4 + sizeof(GeomLds) + 16;
}

// Cross-lane moves as DPP modifiers of VALU instructions (no LDS crossbar round trip).
// quad_perm [1,0,3,2] = 0xB1 (lane ^ 1), [2,3,0,1] = 0x4E (lane ^ 2), row_half_mirror = 0x141
// (lane -> 7 - lane within 8), row_mirror = 0x140 (lane -> 15 - lane within 16), row_shr:n = 0x110 + n,
// row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL>
__device__ inline float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ inline int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
// the value of lane (lane ^ M), M = 1 .. 7: quad permutes, and row_half_mirror (lane ^ 7) in front
// of them for the upper half
template <int M>
__device__ inline int lane_xor(int v) {
  static_assert(M >= 1 && M <= 7, "within aligned groups of eight lanes");
  if (M == 1) return dpp_i<0xB1>(v);
  if (M == 2) return dpp_i<0x4E>(v);
  if (M == 3) return dpp_i<0x1B>(v);
  if (M == 7) return dpp_i<0x141>(v);
  const int h = dpp_i<0x141>(v);             // lane ^ 7, then ^ (M ^ 7)
  return M == 4 ? dpp_i<0x1B>(h) : M == 5 ? dpp_i<0x4E>(h) : dpp_i<0xB1>(h);
}
// min / max over aligned groups of 8 lanes, the result in every lane of the group
__device__ inline float min8(float v) {
  v = fminf(v, dpp_f<0xB1>(v)); v = fminf(v, dpp_f<0x4E>(v)); return fminf(v, dpp_f<0x141>(v));
}
__device__ inline float max8(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); return fmaxf(v, dpp_f<0x141>(v));
}
// inclusive prefix sum over the 64 lanes of a wave (the scan of LLVM's atomic optimizer for GFX9)
__device__ inline int wave_inclusive_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// The geometry of one frame by ONE wave, all 64 lanes: lane = strip * 8 + corner.  Same float32
// operations as strip::frame_geometry (the host's serial version): the corners from
// strip::strip_corners' expressions, rotated as strip::strip_geometry does, their bounding box
// reduced over a strip's eight lanes (min / max of finite values: order independent), the window
// and the two cone edges by the calls the host makes.  Lane 8 s stores strip s.
__device__ inline void frame_geometry_wave64(const strip::RigArgs& c, const StripPose& ps, int lane, GeomLds* g) {
  using namespace strip;
  const int st = lane >> 3, k = lane & 7;
  float ax_lo = 0.0f, ax_hi = 0.0f, tmin = 0.0f, tmax = 0.0f;
#pragma unroll
  for (int q = 0; q < kMaxStrips; ++q) {       // (wave-uniform loads, picked by selects)
    const bool me = st == q;
    ax_lo = me ? c.ax_lo[q] : ax_lo; ax_hi = me ? c.ax_hi[q] : ax_hi;
    tmin = me ? c.tmin[q] : tmin; tmax = me ? c.tmax[q] : tmax;
  }
  const bool live = (st < c.P) & (((c.live_mask >> st) & 1) != 0);
  const float d = (k & 1) ? c.dmax : c.dmin;
  const float ax = (k & 2) ? ax_hi : ax_lo;
  const float gg = (k & 4) ? c.g1 : c.g0;
  float cx = ax * d * c.inv, cz = gg * d * c.inv;                 // strip_corners
  cx = live ? cx : 0.0f; cz = live ? cz : 0.0f;                   // (cfg_rig: a dead strip's corners are zero)
  const Pose p = pose_of(c, ps.y0, ps.y2, ps.y6, ps.y8, ps.tx, ps.tz, ps.wo, ps.ho);
  const float xf = p.y0 * cx + p.y6 * cz + p.xd;                  // strip_geometry
  const float zf = p.y2 * cx + p.y8 * cz + p.zd;
  const float lx = min8(xf), hx = max8(xf), lz = min8(zf), hz = max8(zf);
  const bool on = live & (p.ok != 0);
  bool in = false;
  const Win16 ww = window_of(c, lx, hx, lz, hz, p.slack, in);
  const Line l = cone_edge(c, p, tmin, true), r = cone_edge(c, p, tmax, false);
  const bool inside = on & in & (ww.w > 0);
  const Win16 w = on ? ww : Win16{0, 0, 0, 0};
  if (k == 0) {
    g->win[st] = w;
    g->L[st] = on ? l : Line{0.0f, 0.0f, 0.0f, 0.0f};
    g->R[st] = on ? r : Line{0.0f, 0.0f, 0.0f, 0.0f};
  }
  const unsigned long long in_mask = __builtin_amdgcn_ballot_w64(inside & (k == 0));
  const bool some = w.w > 0;               // (the same in a strip's eight lanes)
  int ux0 = some ? w.x0 : 32767, ux1 = some ? w.x0 + w.w : 0;
  int uz0 = some ? w.z0 : 32767, uz1 = some ? w.z0 + w.h : 0;
  // over the strips: the two strips of a row of 16 lanes by row_mirror, the four rows by readlane
  ux0 = min(ux0, dpp_i<0x140>(ux0)); ux1 = max(ux1, dpp_i<0x140>(ux1));
  uz0 = min(uz0, dpp_i<0x140>(uz0)); uz1 = max(uz1, dpp_i<0x140>(uz1));
  {
    auto rows4 = [](int v, bool take_min) {
      const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
      const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
      return take_min ? min(min(a, b), min(c, d)) : max(max(a, b), max(c, d));
    };
    ux0 = rows4(ux0, true); ux1 = rows4(ux1, false); uz0 = rows4(uz0, true); uz1 = rows4(uz1, false);
  }
  if (lane == 0) {
    g->U = ux1 > ux0 ? Win16{(short)ux0, (short)uz0, (short)(ux1 - ux0), (short)(uz1 - uz0)} : Win16{0, 0, 0, 0};
    g->ok = p.ok;
    g->has_holes = 0;
    int bits = 0;
#pragma unroll
    for (int q = 0; q < kMaxStrips; ++q) bits |= (int)((in_mask >> (8 * q)) & 1ull) << q;
    g->inside = bits;
  }
}

// What a launch leaves in device memory for the kernels behind it (k_strip_combine, the batch
// fuse): written by the scatter kernel's workgroups, never read by them.
struct FrameTables {
  Win16* wins;                // (B, kMaxStrips)
  Win16* unions;              // (B)   union windows, x in whole kSpanAlign groups
  int* counts;                // (B, kMaxStrips)   entries of each strip's shared-group list
  uint32_t* list;             // (B, P, seg_cap)   the groups of a strip's cover nobody owns and no
                              //                   lower strip's cover holds
};

// Status bits the kernels set (dm_status_bits in the header).  Every bit lives in a byte of its
// own and is raised by a plain byte store of 1: two kernels (or two workgroups) raising different
// bits cannot overwrite each other, and no atomic has to cross PCIe to pinned host memory.
constexpr int kStatusFrameDidNotFit = 1, kStatusListOverflow = 0x100;
__device__ inline void raise_status(int* status, int bit) {
  __hip_atomic_store(reinterpret_cast<unsigned char*>(status) + (bit == kStatusFrameDidNotFit ? 0 : 1), (unsigned char)1,
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct StripArgs {
  int W, H;
  int clip, flip_h;
  float cx, cy, fx, fy, res;
  float fx_inv, fy_inv, res_inv;
  float dmin, dmax, hmax;
  float Hm1, mhm1;
  float p4, p5, p7, p8;       // the batch's pitch rotation (Rp[4], Rp[5], Rp[7], Rp[8])
  int wp, P;
  int dc, valid_c;
  int oc, ch0, oc_total;      // channels of this launch's group / first channel / channels of out
  int slab_stride;            // cells per slab = cells of the LDS window region
  int max_rows;               // rows the LDS row tables hold (>= the union window's height)
  int seg_cap;                // entries a strip's shared-group list holds
  float fill;
  int b0;                     // first frame of this launch
  // Where the fill duty of the map rows OUTSIDE the frame's union window (whole rows of fill value:
  // about 60 % of a 512-row map) runs.  0: with everything else under the pixel loop (rounds 2-3).
  // 1: out of the loop, which a working set beyond the Infinity Cache makes bandwidth bound --
  // of every eight such rows of a wave the first `head_share` are stored in the kernel's head
  // (HBM idles there while the row tables are derived), the others by the combine kernel (a chain
  // of dependent round trips under which HBM idles too): kFillSplit* below.
  int defer_outer, head_share;
  int xcd_units;              // value pass: the workgroups of one (frame, strip) share an XCD (its L2 keeps their part of the pixel list)
  const float* poses_dev;     // (B, kPoseFloats) in device memory (prepared frames), or NULL: `poses`
  const float* depth;
  const float* value;         // (B, oc_total, H, W) or NULL: project the heights
  const uint8_t* valid;
  float* slabs;
  float* out;
  uint8_t* mask;
  int mh, mw;
  FrameTables t;
  int* status;                // device-visible status word (or NULL)
  // PLANES (up to four strips, one or two output channels): the groups two or more strips' covers hold are
  // numbered frame-wide -- every workgroup of the frame derives the same numbers from the same covers -- and
  // a strip stores its values of them COMPACTLY, at their numbers, in its plane (B * oc, P, plane_cap) float4
  // (in the slab region); the lowest strip that holds a group also stores {float4 group of the map, covers that
  // hold it} at that number in the frame's segment of the list region (B, plane_cap).  k_strip_combine_planes
  // then needs ONE round of coalesced loads per group.
  int plane_cap;
  uint16_t* list;             // (B, P, H, wp) cells of the pixels inside their strips' windows (index / value pass)
#ifdef DM_STAMPS
  long long* stamps;
#endif
  strip::RigArgs rig;
  StripPose poses[kPoseFrames];     // this launch's frames (b0 ...), unless poses_dev
};

// RED: kMin / kMax.  Always the fast geometry (axis-aligned rotations, exact FMA division),
// 16-byte depth loads.  HAS_VALUE / HAS_VALID / LEAN as in k_window_scatter.
//
// The pixel loop is bound by VALU issue (tools/strip_stamps.py with timing-only builds, tools/experiments: 21 us
// of arithmetic against 15 us of memory traffic at cfg2), so everything a pixel does not need
// is kept out of it: the geometry is derived once in the kernel's head, the ray slope of an image row from
// a table in LDS, the fill duty is wave-level stores with scalar addressing, and a strip whose
// window the map's borders did not clip (FrameGeom::inside) skips the window test -- every
// pixel with a depth in range then lands inside the window by construction.
//
// MODE (value maps of many channels: one index computation per pixel, maps.py:314-350):
//   kProject     the whole projection in one kernel (heights; value maps of few channels).
//   kIndexOut    the index pass: no window, no fill, no flush -- every pixel's cell inside its
//                strip's window (16 bits, 0xffff: rejected) goes to a list in the workspace.
// NT_FILL: the fill duty's float4 stores non-temporal (dm_pixel.hpp kFillCachePolicy) or under the
//   default cache policy (kProject only: the host decides per call, dm_strip.hip nt_fill_pays).
//   kFromList    the value pass, one workgroup per (strip, channel, frame): cells from the list,
//                values from the channel's image -- 8 + 16 bytes per thread and row, four
//                instructions per pixel instead of twenty, four rows in flight.
// (cache policies that were measured and left at the default -- value loads, the flush's map stores, the fill
// duty's mask stores non-temporal: DESIGN 4.6 -- are not switches any more; tools/experiments has the patch)
enum { kProject = 0, kIndexOut = 1, kFromList = 2 };
template <int RED, bool HAS_VALID, bool HAS_VALUE, bool LEAN, int MODE = kProject, bool NT_FILL = true, bool PLANES = false>
__global__ void __launch_bounds__(kScatterThreads)
k_strip_scatter(StripArgs a) {
  constexpr int VEC = 4;
  static_assert(!PLANES || MODE == kProject, "compact planes: the whole projection in one scatter kernel");
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
  // Cache policy of the depth loads: non-temporal in the streaming variant (NT_FILL: the host picks it where
  // the call's bytes cannot stay cache resident anyway, dm_strip.hip streaming_call).  A depth map is read
  // once per call; loaded under the default policy it evicts what IS read again soon (slabs, lists, the
  // cells the batch fuse reads) -- HBM-served cfg2 launch sequence 50.0-50.5 -> 48.1 us, 54.0 -> 49.0 on a
  // slower box (round 4).  Not in the other kernels: the value pass's values measured slower non-temporal
  // (cfg3 1 473 -> 1 545 us), k_strip_fused no different, and the window path's projection is followed by
  // the ego-motion flow kernel, which finds the depth maps in the Infinity Cache only under the default
  // policy (cfg5 flow call 153 -> 176 us).
  constexpr int kDepthPolicy = (NT_FILL && MODE == kProject) ? 2 : 0;
  // fill steps (wave-level, 1 KB each) per pipeline half-iteration: three or four make every wait for
  // the depth loads wait for older stores too (+2 us per extra step at cfg2), none or one moves
  // the stores behind the loop for the same total
  constexpr int kStripFillPerHalf = 2;
  extern __shared__ float lds[];
  // (the value pass runs channel-major: the workgroups of one (frame, strip) -- which read the same
  // part of the pixel list -- are dispatched together)
  int part = MODE == kFromList ? blockIdx.y : blockIdx.x;   // column strip
  int chl = MODE == kFromList ? blockIdx.x : blockIdx.y;    // channel within this launch's group
  int bl = blockIdx.z;
  if (MODE == kFromList && a.xcd_units) {
    // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one: observed, for speed only).  The
    // channels of one (frame, strip) read the same part of the pixel list: with consecutive block indices they
    // sat on all eight XCDs and every L2 fetched that part (the list was read ~14 times per call, PMC).  Here the
    // blocks of one XCD walk the units (frame, strip) it owns channel by channel: one L2 fetches a unit's part once.
    // (Measured: fewer bytes fetched, 2 % more time -- dm_strip.hip g_no_xcd_units; not the default.)
    const int lin = (int)blockIdx.x + (int)gridDim.x * ((int)blockIdx.y + (int)gridDim.y * (int)blockIdx.z);
    const int xcd = lin & 7, slot = lin >> 3, oc = (int)gridDim.x;
    const int round = slot / oc;
    const int unit = round * 8 + xcd;           // (units = strips x frames of the launch: a multiple of 8)
    chl = slot - round * oc;
    bl = unit / a.P;
    part = unit - bl * a.P;
  }
  const int b = a.b0 + bl;
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
      a.list + (MODE != kProject ? ((size_t)b * nparts + part) * (size_t)a.H * a.wp : 0), 0,
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
        const f32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_depth, at << 2, 0, kDepthPolicy);
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

  // the frame's camera state: one batch of scalar loads from the kernel arguments (this launch's
  // frames travel in them) or from the caller's device buffer (prepared frames) -- one uniform
  // pointer into the constant address space either way
  typedef const __attribute__((address_space(4))) float cfloat;
  cfloat* const pr_args = (cfloat*)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                    offsetof(StripArgs, poses)) + bl * kPoseFloats;
  cfloat* const pr = a.poses_dev ? (cfloat*)(a.poses_dev + (size_t)b * kPoseFloats) : pr_args;
#ifdef DM_STAMPS
  long long stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DM_STAMP(0);
  const int table_off = a.slab_stride + 64;
  const int P2 = strips_pow2(nparts);
  uint32_t* covers = reinterpret_cast<uint32_t*>(lds + table_off);                       // [row of U][strip], row stride P2
  uint32_t* owned_t = covers + P2 * a.max_rows;                                          // this strip's owned span of every row of U
  uint2* reach = reinterpret_cast<uint2*>(lds + table_off + (P2 + 1) * a.max_rows);       // {lo, width} of every row of U
  int* list_at = reinterpret_cast<int*>(lds + table_off + (P2 + 3) * a.max_rows);        // list entries of this strip before the row, within its chunk of 64 rows
  float* aytab = lds + table_off + (P2 + 4) * a.max_rows;
  GeomLds* geom = reinterpret_cast<GeomLds*>(aytab + ((a.H + 3) & ~3));
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lane = (int)threadIdx.x & 63;
  if (wave == 0) {
    // The frame's geometry: wave 0 derives it (lane = strip x corner) into LDS, from the start of
    // the kernel, while the other fifteen waves initialise the window and the ray-slope table;
    // everybody reads it behind the barrier.
    // (the record in the kernel arguments is requested with the arguments themselves -- its address
    // depends on nothing that has to be loaded first; a prepared batch's record in device memory
    // takes one more round trip)
    StripPose ps;
    ps.y0 = pr_args[0]; ps.y2 = pr_args[1]; ps.y6 = pr_args[2]; ps.y8 = pr_args[3]; ps.tx = pr_args[4];
    ps.tz = pr_args[5]; ps.wo = pr_args[6]; ps.ho = pr_args[7]; ps.cam_h = pr_args[8];
    if (a.poses_dev) {
      ps.y0 = pr[0]; ps.y2 = pr[1]; ps.y6 = pr[2]; ps.y8 = pr[3]; ps.tx = pr[4]; ps.tz = pr[5]; ps.wo = pr[6]; ps.ho = pr[7];
      ps.cam_h = pr[8];
    }
    ps.pad0 = ps.pad1 = ps.pad2 = 0.0f;
#ifdef DM_STAMPS
    asm volatile("" : "+s"(ps.y0), "+s"(ps.y2), "+s"(ps.cam_h));
#endif
    DM_STAMP(9);
    frame_geometry_wave64(a.rig, ps, lane, geom);
    DM_STAMP(10);
  } else {
    // The whole window region of LDS gets the fill value, the ray-slope table its H entries
    // (maps.py:670-678; border rows poisoned), under the first depth rows in flight.
    constexpr int kInitThreads = kScatterThreads - 64;
    const int t = (int)threadIdx.x - 64;
    const float lds_init = kDeferCamH ? (RED == kMax ? -INFINITY : INFINITY) : a.fill;
    if (MODE != kIndexOut)
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
  DM_STAMP(7);
  Window U = widen(geom->U);
  {   // the union window in whole span groups (covers reach that far)
    const int ux1 = min((U.x0 + U.w + strip::kSpanAlign - 1) & ~(strip::kSpanAlign - 1), a.mw);
    U.x0 &= ~(strip::kSpanAlign - 1);
    U.w = U.w > 0 ? ux1 - U.x0 : 0;
  }
  // A frame that does not fit what the host sized the launch for (cannot happen when the host
  // derived the sizes from these very frames: a caller's device buffer changed behind the
  // plan's back): flag it and project nothing -- every cell of its maps gets the fill value.
  bool fits = geom->ok != 0 && U.h <= a.max_rows && U.h <= kListMaxRows && U.w <= 4 * kListMaxGroups;
  {
    const Win16 mine = geom->win[lane & (strip::kMaxStrips - 1)];
    fits = fits && __builtin_amdgcn_ballot_w64((int)mine.w * mine.h > a.slab_stride) == 0;
  }
  Window w = widen(geom->win[part]);
  int flags = geom->inside;
  {
    // wave-uniform values: keep them in SGPRs
    const int fit = __builtin_amdgcn_readfirstlane((int)fits);
    w.x0 = __builtin_amdgcn_readfirstlane(w.x0); w.z0 = __builtin_amdgcn_readfirstlane(w.z0);
    w.w = __builtin_amdgcn_readfirstlane(fit ? w.w : 0); w.h = __builtin_amdgcn_readfirstlane(fit ? w.h : 0);
    U.x0 = __builtin_amdgcn_readfirstlane(U.x0); U.z0 = __builtin_amdgcn_readfirstlane(U.z0);
    U.w = __builtin_amdgcn_readfirstlane(fit ? U.w : 0); U.h = __builtin_amdgcn_readfirstlane(fit ? U.h : 0);
    flags = __builtin_amdgcn_readfirstlane(fit ? flags : 0);
    fits = fit != 0;
  }
  const int area = w.w * w.h;
  // what the kernels behind this one read of the frame: by the first strip's workgroup of the
  // launch's first channel
  if (part == 0 && chl == 0 && wave == 0) {
    if (lane < strip::kMaxStrips) a.t.wins[(size_t)b * strip::kMaxStrips + lane] = fits ? geom->win[lane] : Win16{0, 0, 0, 0};
    if (lane == 0) {
      a.t.unions[b] = narrow16(U);
      if (!fits && a.status) raise_status(a.status, kStatusFrameDidNotFit);
    }
  }
  // the maps of this (frame, channel) as raw buffer resources (the fill duty's and the flush's stores)
  const size_t map_base = ((size_t)b * a.oc_total + ch) * (size_t)a.mh * a.mw;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const unsigned map_cells = (unsigned)a.mh * (unsigned)a.mw;
  const __amdgpu_buffer_rsrc_t rs_out =
      __builtin_amdgcn_make_buffer_rsrc(a.out + map_base, 0, a.out ? map_cells * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_mask =
      __builtin_amdgcn_make_buffer_rsrc(a.mask + map_base, 0, a.out ? map_cells : 0u, 0x00020000);
  const unsigned fill_bits = __float_as_uint(a.fill);
  // The row tables, one thread per row of the union window (whole waves: the rows of a wave are a
  // chunk of 64): the cover of every strip on the row (strip::row_cover), this strip's owned
  // span (strip::row_owned: its cover cut by the others'), the hull of the covers ("reach":
  // outside it the map holds the fill value), and where the row's entries of the strip's
  // shared-group list go -- a strip lists the groups of its cover that it does not own, in x
  // order, so a row's entries follow those of the rows before it: a scan over the chunk (here)
  // and over the chunks (wave 0, behind the barrier).  Straight-line code over 4 or 8 strips, the
  // covers in registers (a dead strip has an empty window and an empty cover).
  auto row_tables = [&](auto p2_tag) {
    constexpr int kP2 = decltype(p2_tag)::value;
    for (int r0 = wave * 64; r0 < U.h; r0 += kScatterThreads) {       // (wave-uniform trips: the scan's shuffles)
      const int r = r0 + lane;
      const bool live = r < U.h;
      uint32_t cov[kP2];
      int rlo = 32767, rhi = 0;
#pragma unroll
      for (int q = 0; q < kP2; ++q) {
        // (the strips' windows and edges: broadcast reads of the geometry in LDS)
        const Win16 gw = geom->win[q];
        const strip::Line gl = geom->L[q], gr = geom->R[q];
        cov[q] = strip::row_cover(gw, gl, gr, U.z0 + r, a.mw);
        rlo = min(rlo, cov[q] ? (int)(cov[q] & 0xffffu) : 32767); rhi = max(rhi, (int)(cov[q] >> 16));
      }
      uint32_t owned = 0u;
      int entries;
      bool hole = false;
      if (PLANES && kP2 == 4) {
        // compact planes: the row's cells in two or more covers -- the same number in every workgroup of the frame
        // (the numbering of the shared groups) -- and the cells in any cover: fewer than the hull holds = a hole
        int covered = 0;
        entries = strip::cover_measures(cov[0], cov[1 % kP2], cov[2 % kP2], cov[3 % kP2], 1 << 30, &covered);
        hole = rhi > rlo && covered < rhi - rlo;
      } else {
        // this strip's cover and its owned span: the cover cut by the others' in strip::row_owned's
        // order (part ^ 1, part ^ 2, ...), picked with selects (part is wave-uniform)
        uint32_t mine = cov[0];
#pragma unroll
        for (int q = 1; q < kP2; ++q) mine = part == q ? cov[q] : mine;
        int lo = (int)(mine & 0xffffu), hi = (int)(mine >> 16);
#pragma unroll
        for (int m = 1; m < kP2; ++m) {
          const int o = part ^ m;
          uint32_t other = cov[0];
#pragma unroll
          for (int q = 1; q < kP2; ++q) other = o == q ? cov[q] : other;
          strip::cut_span(lo, hi, other);
        }
        owned = hi > lo ? (uint32_t)lo | ((uint32_t)hi << 16) : 0u;
        // entries of this strip's list on the row: the groups of its cover outside the owned span
        entries = (int)((mine >> 16) - (mine & 0xffffu)) - (hi > lo ? hi - lo : 0);
        // A hole between covers inside the hull: some cover ends inside the hull where no other cover
        // continues.
#pragma unroll
        for (int q = 0; q < kP2; ++q) {
          const int end = (int)(cov[q] >> 16);
          bool continued = (cov[q] == 0u) | (end >= rhi);
#pragma unroll
          for (int o = 0; o < kP2; ++o)
            if (o != q) continued = continued | (((int)(cov[o] & 0xffffu) <= end) & (end < (int)(cov[o] >> 16)));
          hole = hole | !continued;
        }
      }
      entries = live ? entries >> 2 : 0;
      const int before = wave_inclusive_scan(entries);       // over the wave's rows
      if (lane == 63) geom->chunk_entries[r0 >> 6] = before;
      if (live) {
        if (kP2 == 4) {
          *reinterpret_cast<uint4*>(covers + r * 4) = make_uint4(cov[0], cov[1], cov[2], cov[3]);
        } else {
          *reinterpret_cast<uint4*>(covers + r * 8) = make_uint4(cov[0], cov[1], cov[2], cov[3]);
          *reinterpret_cast<uint4*>(covers + r * 8 + 4) = make_uint4(cov[4 % kP2], cov[5 % kP2], cov[6 % kP2], cov[7 % kP2]);
        }
        if (!PLANES) owned_t[r] = owned;
        reach[r] = rhi > rlo ? make_uint2((unsigned)rlo, (unsigned)(rhi - rlo)) : make_uint2(0u, 0u);
        list_at[r] = before - entries;
        // (groups of the hull in no strip's cover: stored behind the barrier by the workgroup whose fill
        // duty owns the map row -- flagged here, so that the others skip that pass)
        if (__builtin_expect(hole, 0) && (U.z0 + r) % nparts == part) geom->has_holes = 1;
      }
    }
  };
  // More than four strips (small frames, narrow windows): eight lanes per row (lane = row x strip;
  // all 1024 threads busy on 128 rows of the union window at a time) -- the straight-line version
  // above with eight covers per thread does not fit the kernel's registers, and rolled loops over
  // covers in LDS took 6.3 us at the reference demo's frame (320x240, eight strips) against 2.7 us
  // this way (at four strips the version above is the faster one, 1.8 against 2.2 us: it needs no
  // second pass).  Each lane computes ONE cover, sees the other strips' through lane-xor DPP moves -- in
  // strip::row_owned's order, so every lane cuts its own cover to its owned span on the way -- and
  // answers "does another cover continue where mine ends?" for the hole test; the lane of this
  // workgroup's strip stores the owned span and the row's number of list entries, lane 0 of the
  // row the reach.  Then (behind a barrier) the scan over the rows, one lane per row.
  auto row_tables_lanes = [&](auto p2_tag) {
    constexpr int kP2 = decltype(p2_tag)::value;
    constexpr int kShift = kP2 == 4 ? 2 : 3;
    constexpr int kRowsPerPass = kScatterThreads >> kShift;
    const int q = lane & (kP2 - 1);
    const Win16 gw = geom->win[q];
    const strip::Line gl = geom->L[q], gr = geom->R[q];
    for (int r0 = 0; r0 < U.h; r0 += kRowsPerPass) {       // (uniform trips: the DPP moves)
      const int r = r0 + ((int)threadIdx.x >> kShift);
      const bool live = r < U.h;
      const uint32_t cov = live ? strip::row_cover(gw, gl, gr, U.z0 + r, a.mw) : 0u;
      const int end = (int)(cov >> 16);
      int rlo = cov ? (int)(cov & 0xffffu) : 32767, rhi = end;
      int lo = (int)(cov & 0xffffu), hi = end;
      bool continued = cov == 0u;
      auto see = [&](int m, uint32_t other) {
        rlo = min(rlo, other ? (int)(other & 0xffffu) : 32767); rhi = max(rhi, (int)(other >> 16));
        strip::cut_span(lo, hi, other);
        continued = continued | (((int)(other & 0xffffu) <= end) & (end < (int)(other >> 16)));
      };
      see(1, (uint32_t)lane_xor<1>((int)cov)); see(2, (uint32_t)lane_xor<2>((int)cov)); see(3, (uint32_t)lane_xor<3>((int)cov));
      if (kP2 == 8) {
        see(4 % kP2, (uint32_t)lane_xor<4>((int)cov)); see(5 % kP2, (uint32_t)lane_xor<5>((int)cov));
        see(6 % kP2, (uint32_t)lane_xor<6>((int)cov)); see(7 % kP2, (uint32_t)lane_xor<7>((int)cov));
      }
      continued = continued | (end >= rhi);
      // a hole between covers inside the hull: some cover ends inside the hull where no other continues
      const unsigned long long open = __builtin_amdgcn_ballot_w64(!continued);
      const bool hole = ((open >> (lane & ~(kP2 - 1))) & ((1ull << kP2) - 1ull)) != 0ull;
      if (live) {
        covers[r * kP2 + q] = cov;
        if (q == part) {
          owned_t[r] = hi > lo ? (uint32_t)lo | ((uint32_t)hi << 16) : 0u;
          // entries of this strip's list on the row: the groups of its cover outside the owned span
          list_at[r] = ((int)((cov >> 16) - (cov & 0xffffu)) - (hi > lo ? hi - lo : 0)) >> 2;
        }
        if (q == 0) {
          reach[r] = rhi > rlo ? make_uint2((unsigned)rlo, (unsigned)(rhi - rlo)) : make_uint2(0u, 0u);
          if (__builtin_expect(hole, 0) && (U.z0 + r) % nparts == part) geom->has_holes = 1;      // (see row_tables)
        }
      }
    }
    lds_barrier();
    // where the row's entries of the strip's list go: a scan over chunks of 64 rows here, over
    // the chunks by wave 0 behind the next barrier
    for (int r0 = wave * 64; r0 < U.h; r0 += kScatterThreads) {       // (wave-uniform trips: the scan's shuffles)
      const int r = r0 + lane;
      const bool live = r < U.h;
      const int entries = live ? list_at[r] : 0;
      const int before = wave_inclusive_scan(entries);
      if (lane == 63) geom->chunk_entries[r0 >> 6] = before;
      if (live) list_at[r] = before - entries;
    }
  };
  if (P2 == 4) row_tables(std::integral_constant<int, 4>{}); else row_tables_lanes(std::integral_constant<int, 8>{});
  // Fill duty, head share: of this wave's map rows (part + (wave + 16 j) P) those outside the union
  // window's rows hold nothing but the fill value -- known from the geometry alone -- and those
  // with (j & 7) < head_share are stored right here: the waves that derive no row tables have
  // nothing to do until the barrier, and HBM has nothing to do either.  Whole rows, wave-level
  // stores with scalar addressing (as the loop's fill steps).
  if (MODE != kIndexOut && a.out != nullptr && a.defer_outer && a.head_share > 0) {
    const int rows_mine = (a.mh - part + nparts - 1) / nparts;
    const int nj = wave < rows_mine ? (rows_mine - wave + 15) >> 4 : 0;
    const int chunks_h = (a.mw + 255) >> 8;
    for (int j = 0; j < nj; ++j) {
      const int z = part + (wave + 16 * j) * nparts;
      if ((unsigned)(z - U.z0) < (unsigned)U.h || (j & 7) >= a.head_share) continue;      // (scalar)
      for (int c = 0; c < chunks_h; ++c) {
        const int x = (c << 8) + (lane << 2);
        const int cell0 = z * a.mw + (c << 8);
        buffer_store_b128_at_scalar_offset<NT_FILL ? kFillCachePolicy : 0>(
            (u32x4){fill_bits, fill_bits, fill_bits, fill_bits}, rs_out, x < a.mw ? lane << 4 : 0x7ffffff0, cell0 << 2);
        __builtin_amdgcn_raw_buffer_store_b32(0u, rs_mask, x < a.mw ? lane << 2 : 0x7ffffff0, cell0, 0);
      }
    }
  }
  DM_STAMP(8);
  // What the pixel loop reads of the kernel arguments and of the frame's pose is (re)loaded HERE,
  // through pointers the compiler cannot see through: a scalar whose live range crossed the
  // geometry code above would be spilled as a whole and reloaded at every use inside the loop.
  typedef const __attribute__((address_space(4))) StripArgs cargs;
  cargs* la = (cargs*)__builtin_amdgcn_kernarg_segment_ptr();
  cfloat* pl = pr;
  asm volatile("" : "+s"(la), "+s"(pl));
  float fy0 = pl[0], fy2 = pl[1], fy6 = pl[2], fy8 = pl[3], ftx = pl[4], ftz = pl[5];
  float wo = pl[6], ho = pl[7], cam_h = pl[8];
  float p4 = la->p4, p5 = la->p5, p7 = la->p7, p8 = la->p8;
  // (one batch of scalar loads, pinned)
  asm volatile("" : "+s"(p4), "+s"(p5), "+s"(p7), "+s"(p8), "+s"(cam_h), "+s"(fy0), "+s"(fy2),
                    "+s"(fy6), "+s"(fy8), "+s"(ftx), "+s"(ftz), "+s"(wo), "+s"(ho));
  lds_barrier();
  DM_STAMP(2);
  // Groups of a row's hull that lie in NO strip's cover hold the fill value, and neither the fill duty
  // (outside the hull) nor a flush (inside a cover) writes them.  They occur where one strip's window has
  // ended between two others' -- a few rows of one or two frames in a batch of random poses.  The
  // workgroup whose fill duty owns the map row stores them here, wave-level: wave v takes the rows
  // part + (v + 16 j) P inside the union window, 256 cells of the hull per step, a lane stores its
  // group when no cover holds it; the row tables flag the workgroups that have such rows, the others
  // skip the pass (unconditionally it cost every workgroup 1.5 us).  Until round 4 ONE thread per such row stored them group by group inside the
  // row tables: 6.5 us in the head of the workgroups concerned -- and every launch of the batch waited
  // for them (profiles/r04_scatter_phase_stamps.log).
  if (MODE != kIndexOut && __builtin_expect(__builtin_amdgcn_readfirstlane(geom->has_holes) != 0, 0)) {
    const int first = U.z0 + ((part - U.z0 % nparts + nparts) % nparts);      // first row of the union window that is this workgroup's
    for (int z = first + wave * nparts; z < U.z0 + U.h; z += 16 * nparts) {  // (scalar)
      const int r = z - U.z0;
      const uint2 hull = reach[r];
      const uint4 c0 = *reinterpret_cast<const uint4*>(covers + r * P2);
      uint4 c1 = make_uint4(0u, 0u, 0u, 0u);
      if (P2 == 8) c1 = *reinterpret_cast<const uint4*>(covers + r * 8 + 4);
      for (unsigned x0 = hull.x; x0 < hull.x + hull.y; x0 += 256) {          // (scalar: the hull is the row's, the same in every lane)
        const int x = (int)x0 + (lane << 2);
        const int covered = (int)strip::in_span(c0.x, x) | (int)strip::in_span(c0.y, x) | (int)strip::in_span(c0.z, x) |
                            (int)strip::in_span(c0.w, x) | (int)strip::in_span(c1.x, x) | (int)strip::in_span(c1.y, x) |
                            (int)strip::in_span(c1.z, x) | (int)strip::in_span(c1.w, x);
        const bool store = covered ? false : x < (int)(hull.x + hull.y);
        const int cell0 = z * a.mw + (int)x0;
        buffer_store_b128_at_scalar_offset<NT_FILL ? kFillCachePolicy : 0>(
            (u32x4){fill_bits, fill_bits, fill_bits, fill_bits}, rs_out, store ? lane << 4 : 0x7ffffff0, cell0 << 2);
        __builtin_amdgcn_raw_buffer_store_b32(0u, rs_mask, store ? lane << 2 : 0x7ffffff0, cell0, 0);
      }
    }
  }
  if (wave == 0) {       // list entries before each chunk of 64 rows (the flush reads them behind the next barrier)
    const int chunks_u = (U.h + 63) >> 6;                      // <= kListMaxRows / 64 = 64
    const int mine = lane < chunks_u ? geom->chunk_entries[lane] : 0;
    const int before = wave_inclusive_scan(mine);
    geom->chunk_entries[lane] = before - mine;
    if (lane == 63) geom->chunk_entries[kListMaxRows / 64] = before;
  }
  DM_STAMP(3);
  // Fill duty: map rows part, part + P, ... of (b, ch) outside the rows' reach spans (the hull of
  // the strips' covers: inside it the flush below and the combine step write).  Wave-level: wave v takes the rows part + (v + 16 j) F,
  // one step stores 256 cells of a row (float4 per lane) and their mask bytes with SCALAR
  // addressing -- the map of this (frame, channel) as a raw buffer resource, the row and chunk
  // in the scalar offset, the lane's fixed 16 / 4 bytes in the vector offset.  A lane that has
  // nothing to write (inside the reach, past the row's end, past the wave's rows) gets a vector offset
  // past the end of the buffer and is dropped by the hardware's range check: no branch in the
  // loop, a handful of VALU instructions per KB.
  const int fparts = nparts;
  const int fill_rows = (la->mh - part + fparts - 1) / fparts;
  const int chunks = (la->mw + 255) >> 8;
  // this wave's rows part + (wave + 16 j) P, j in [j_lo, j_hi): all of them, or -- when the rows
  // outside the union window are stored elsewhere (defer_outer) -- those inside its z range
  int j_lo = 0, j_hi = wave < fill_rows ? (fill_rows - wave + 15) >> 4 : 0;
  if (la->defer_outer) {
    const int z_base = part + wave * fparts, stride = 16 * fparts;
    const int a0 = U.z0 - z_base + stride - 1, a1 = U.z0 + U.h - z_base + stride - 1;
    const int lo = a0 > 0 ? a0 / stride : 0, hi = a1 > 0 ? a1 / stride : 0;
    j_lo = __builtin_amdgcn_readfirstlane(U.h > 0 ? min(lo, j_hi) : 0);
    j_hi = __builtin_amdgcn_readfirstlane(U.h > 0 ? min(hi, j_hi) : 0);
  }
  const int fill_steps = la->out != nullptr && j_hi > j_lo ? (j_hi - j_lo) * chunks : 0;
  const int lane4 = lane << 2;
  int fs = 0, f_row = part + (wave + 16 * j_lo) * fparts, f_chunk = 0;       // (wave-uniform)
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
    const bool skip = !live | (x >= la->mw) | ((unsigned)(x - (int)f_reach.x) < f_reach.y);
    // (the scalar offset must be the same in every lane, skipping or not: a lane-dependent one
    // costs a waterfall loop per store)
    const int cell0 = __builtin_amdgcn_readfirstlane(live ? f_row * la->mw + (f_chunk << 8) : 0);
    buffer_store_b128_at_scalar_offset<NT_FILL ? kFillCachePolicy : 0>(
        (u32x4){fill_bits, fill_bits, fill_bits, fill_bits}, rs_out, skip ? 0x7ffffff0 : lane4 << 2, cell0 << 2);
    __builtin_amdgcn_raw_buffer_store_b32(0u, rs_mask, skip ? 0x7ffffff0 : lane4, cell0, 0);
    ++fs;
    advance(f_row, f_chunk);
    f_reach = f_reach1;
    f_reach1 = reach_of(f_row2);
    advance(f_row2, f_chunk2);
  };
  // (a local map's records carry a neutral yaw and no translation: dm_strip.hip stage_frames)
  const float y0 = fy0, y2r = fy2, y6 = fy6, y8 = fy8, tx = ftx, tz = ftz;
  const float flip_s = la->flip_h ? -1.0f : 1.0f, flip_c = la->flip_h ? la->mhm1 : 0.0f;
  // LDS addresses of the pixel loop as plain 32-bit byte addresses (an LDS pointer IS that), so
  // that the window's origin and the base of `lds` fold into one scalar:
  // address = (z * w + x) * 4 + origin
  typedef __attribute__((address_space(3))) float lds_float;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_float*)lds;
  const unsigned dummy = lds_base + (((unsigned)la->slab_stride + (unsigned)lane) << 2);   // 64 scratch cells
  const int origin = (int)lds_base - 4 * (w.z0 * w.w + w.x0);
  auto lds_at = [](unsigned addr) { return (lds_float*)(uintptr_t)addr; };

  if (area > 0) {
    for (int g = gx; g < nx; g += ntx) {       // one trip unless the strip is wider than the block
      const int q = q0 + g * VEC;
      float ax[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float d = (float)(q + k) - la->cx;
        ax[k] = div_markstein(d, la->fx, la->fx_inv);
        if (!LEAN) ax[k] = (q + k < la->clip || q + k >= la->W - la->clip) ? qnan : ax[k];
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
#pragma unroll
            for (int k = 0; k < VEC; ++k) lds_reduce<RED>(lds_at(li[k]), hv[k]);
          }
          return;
        }
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
            const f2 ri = {la->res_inv, la->res_inv}, nres = {-la->res, -la->res};
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
            bool ok = (zz >= la->dmin) & (zz <= la->dmax);
            if (kTest) ok = ok & ((unsigned)(ix - w.x0) < (unsigned)w.w) & ((unsigned)(iz - w.z0) < (unsigned)w.h);
            if (!LEAN) ok = ok & !__builtin_isunordered(xf, zf) & (h1 <= la->hmax);
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
                const bool ok = (z[u][k] >= la->dmin) & (z[u][k] <= la->dmax);
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
                               :: "s"(la->dmin), "s"(la->dmax), "v"(z[u][k]), "v"(li[k]), "v"(hv[k]), "s"(exec_all)
                               : "vcc", "memory");
                else
                  asm volatile("v_cmpx_le_f32_e32 vcc, %0, %2\n\tv_cmpx_ge_f32_e32 vcc, %1, %2\n\t"
                               "ds_min_f32 %3, %4\n\ts_mov_b64 exec, %5"
                               :: "s"(la->dmin), "s"(la->dmax), "v"(z[u][k]), "v"(li[k]), "v"(hv[k]), "s"(exec_all)
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
                                                  (__mul24(rr, la->wp) + (q - q0)) << 1, 0, 0);
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
        if (wave >= 12) __builtin_amdgcn_s_setprio(3);
        else if (wave >= 8) __builtin_amdgcn_s_setprio(2);
        else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
        for (int it = 0; it < niter; it += 2) {
#pragma unroll
          for (int t = 0; t < kStripFillPerHalf; ++t) fill_step();
          project_rows(tested, za, va, aya, r);
          if (it + 1 < niter) {
            load_rows(za, va, r + 2 * step);
            load_ay(aya, r + 2 * step);
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
  // combines those with the other strips'.  The strip also LISTS such a group for the combine
  // kernel when no lower strip's cover holds it (so every shared group is listed exactly once;
  // a group some strip owns lies in no other strip's cover and is never listed): an entry with
  // the covers that hold the group, appended to this strip's segment of the frame's list.  The
  // index pass (no window) only lists; the value pass (kFromList) only flushes.
  // What only the flush needs is read from the kernel arguments AGAIN, behind the loop, through a
  // pointer the compiler cannot see through: kept live across the pixel loop these values (and the
  // addresses derived from them) cost SGPR spills inside it.
  cargs* fa = (cargs*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(fa));
  const int fP2 = strips_pow2(fa->P);
  const uint32_t* const fcovers = reinterpret_cast<const uint32_t*>(lds + fa->slab_stride + 64);
  const uint32_t* const fowned = fcovers + fP2 * fa->max_rows;
  const int* const flist_at = reinterpret_cast<const int*>(fcovers + (fP2 + 3) * fa->max_rows);
  const GeomLds* const fgeom = reinterpret_cast<const GeomLds*>(
      reinterpret_cast<const float*>(fcovers) + (fP2 + 4) * fa->max_rows + ((fa->H + 3) & ~3));
  const bool emit = MODE != kFromList && chl == 0;
  const int unit = b * fa->oc + chl;               // (frame, channel): the P workgroups that share a map
  auto mask_bits = [&](float4 v) {
    return (uint32_t)mask_of(v.x, fa->fill) | ((uint32_t)mask_of(v.y, fa->fill) << 8) |
           ((uint32_t)mask_of(v.z, fa->fill) << 16) | ((uint32_t)mask_of(v.w, fa->fill) << 24);
  };
  if (!PLANES && area > 0 && (MODE != kIndexOut || emit)) {
    const int l16 = (int)threadIdx.x & 15;
    float* slab = fa->slabs + ((size_t)unit * nparts + part) * fa->slab_stride;
    uint32_t* seg = fa->t.list + ((size_t)b * nparts + part) * fa->seg_cap;
    const uint32_t lower = (1u << part) - 1u;
    for (int row = (int)threadIdx.x >> 4; row < w.h; row += kScatterThreads / 16) {
      const int z = w.z0 + row;
      const int ur = z - U.z0;
      const uint32_t cover = fcovers[ur * fP2 + part], owned = fowned[ur];
      const int lo = (int)(cover & 0xffffu), hi = (int)(cover >> 16);
      const int olo = owned ? (int)(owned & 0xffffu) : lo, ohi = owned ? (int)(owned >> 16) : lo;
      const int cell0 = row * w.w - w.x0;
      // where the row's list entries start: those of the rows before it in its chunk + of the chunks before
      const int at0 = emit ? flist_at[ur] + fgeom->chunk_entries[ur >> 6] : 0;
      for (int x = lo + (l16 << 2); x < hi; x += 64) {
        const bool mine = (x >= olo) & (x < ohi);
        if (MODE != kIndexOut) {
          float4 v = *reinterpret_cast<const float4*>(lds + cell0 + x);
          if (kDeferCamH) {
            v.x = combine<RED>(v.x + cam_h, fa->fill); v.y = combine<RED>(v.y + cam_h, fa->fill);
            v.z = combine<RED>(v.z + cam_h, fa->fill); v.w = combine<RED>(v.w + cam_h, fa->fill);
          }
          if (mine) {
            const int cell = z * fa->mw + x;
            __builtin_amdgcn_raw_buffer_store_b128((f32x4){v.x, v.y, v.z, v.w}, rs_out, cell << 2, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(mask_bits(v), rs_mask, cell, 0, 0);
          } else {
            *reinterpret_cast<float4*>(slab + cell0 + x) = v;
          }
        }
        if (emit && !mine) {       // (the lanes with a group the strip does not own)
          // the covers that hold the group; the entry is live (hits != 0) only where no lower
          // strip's cover does -- every shared group is combined exactly once
          const uint4 c0 = *reinterpret_cast<const uint4*>(fcovers + ur * fP2);
          uint32_t hits = (strip::in_span(c0.x, x) ? 1u : 0u) | (strip::in_span(c0.y, x) ? 2u : 0u) |
                          (strip::in_span(c0.z, x) ? 4u : 0u) | (strip::in_span(c0.w, x) ? 8u : 0u);
          if (fP2 == 8) {
            const uint4 c1 = *reinterpret_cast<const uint4*>(fcovers + ur * 8 + 4);
            hits |= (strip::in_span(c1.x, x) ? 16u : 0u) | (strip::in_span(c1.y, x) ? 32u : 0u) |
                    (strip::in_span(c1.z, x) ? 64u : 0u) | (strip::in_span(c1.w, x) ? 128u : 0u);
          }
          hits = (hits & lower) == 0u ? hits : 0u;
          // its place: the groups of the cover left of the owned span come first, then those right of it
          const int at = at0 + ((x < olo ? x - lo : (olo - lo) + (x - ohi)) >> 2);
          if (at < fa->seg_cap) seg[at] = pack_shared(ur, (x - U.x0) >> 2, hits);
        }
      }
    }
  }
  if (!PLANES && emit && threadIdx.x == 0) {
    const int n = area > 0 ? fgeom->chunk_entries[kListMaxRows / 64] : 0;
    fa->t.counts[(size_t)b * strip::kMaxStrips + part] = n < fa->seg_cap ? n : fa->seg_cap;
    if (n > fa->seg_cap && fa->status)       // (cannot happen: a strip lists at most the groups of its window)
      raise_status(fa->status, kStatusListOverflow);
  }
  if (PLANES) {
    // Compact planes (at most four strips).  A group of this strip's cover that NO other strip's cover holds goes
    // straight to the map; a group two or more covers hold has a number in the frame -- the shared groups of the
    // rows before it (the row tables' scan: the same in every workgroup of the frame) + those of its row left of
    // it (strip::shared_before) -- and the strip stores its values at that number in its plane; the lowest strip
    // that holds the group also stores where it lies in the map and which covers hold it.
    const int cap = fa->plane_cap;
    const __amdgpu_buffer_rsrc_t rs_plane = __builtin_amdgcn_make_buffer_rsrc(
        fa->slabs + ((size_t)unit * nparts + part) * (size_t)cap * 4, 0, (unsigned)cap * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_meta = __builtin_amdgcn_make_buffer_rsrc(
        fa->t.list + (size_t)b * cap, 0, (unsigned)cap * 4u, 0x00020000);
    if (area > 0) {
      const int l16 = (int)threadIdx.x & 15;
      const uint32_t me = 1u << part, lower = me - 1u;
      for (int row = (int)threadIdx.x >> 4; row < w.h; row += kScatterThreads / 16) {
        const int z = w.z0 + row;
        const int ur = z - U.z0;
        const uint4 c0 = *reinterpret_cast<const uint4*>(fcovers + ur * 4);
        const uint32_t cover = part == 0 ? c0.x : part == 1 ? c0.y : part == 2 ? c0.z : c0.w;
        const int lo = (int)(cover & 0xffffu), hi = (int)(cover >> 16);
        const int cell0 = row * w.w - w.x0;
        const int base = flist_at[ur] + fgeom->chunk_entries[ur >> 6];
        for (int x = lo + (l16 << 2); x < hi; x += 64) {
          float4 v = *reinterpret_cast<const float4*>(lds + cell0 + x);
          if (kDeferCamH) {
            v.x = combine<RED>(v.x + cam_h, fa->fill); v.y = combine<RED>(v.y + cam_h, fa->fill);
            v.z = combine<RED>(v.z + cam_h, fa->fill); v.w = combine<RED>(v.w + cam_h, fa->fill);
          }
          const uint32_t hits = (strip::in_span(c0.x, x) ? 1u : 0u) | (strip::in_span(c0.y, x) ? 2u : 0u) |
                                (strip::in_span(c0.z, x) ? 4u : 0u) | (strip::in_span(c0.w, x) ? 8u : 0u);
          const int cell = z * fa->mw + x;
          if (hits == me) {
            __builtin_amdgcn_raw_buffer_store_b128((f32x4){v.x, v.y, v.z, v.w}, rs_out, cell << 2, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(mask_bits(v), rs_mask, cell, 0, 0);
          } else {
            const int pos = base + (strip::shared_before(c0.x, c0.y, c0.z, c0.w, x) >> 2);
            // (a number past the planes' end cannot occur -- the host sizes them for every group of every window --
            // and is dropped by the buffers' range check)
            __builtin_amdgcn_raw_buffer_store_b128((f32x4){v.x, v.y, v.z, v.w}, rs_plane, pos < cap ? pos << 4 : 0x7ffffff0, 0, 0);
            if (emit && (hits & lower) == 0u)
              __builtin_amdgcn_raw_buffer_store_b32((uint32_t)(cell >> 2) | (hits << 28), rs_meta, pos < cap ? pos << 2 : 0x7ffffff0, 0, 0);
          }
        }
      }
    }
    if (emit && part == 0 && threadIdx.x == 0) {      // the frame's shared groups (every workgroup of the frame has the number)
      const int n = fgeom->chunk_entries[kListMaxRows / 64];
      fa->t.counts[(size_t)b * strip::kMaxStrips] = n < cap ? n : cap;
      if (n > cap && fa->status) raise_status(fa->status, kStatusListOverflow);
    }
  }
  DM_STAMP(6);
  DM_STAMPS_OUT();
}

struct StripCombineArgs {
  int b0, oc, ch0, oc_total, mh, mw;
  int P, slab_stride, seg_cap;
  int defer_outer, head_share;      // StripArgs': the rows outside the union window this kernel fills
  float fill;
  const Win16* g_wins;
  const Win16* g_unions;
  const int* g_counts;        // (B, kMaxStrips)
  const uint32_t* g_list;     // (B, P, seg_cap)
  const float* slabs;
  float* out;
  uint8_t* mask;
};

constexpr int kCombineThreads = 256;
// A (frame, channel) gets kCombineSlots / ENTRIES blocks; block x works on the list segment of strip
// x mod P (the strips list their shared groups themselves, each into its own segment), its
// threads take ENTRIES entries at a time (their slab loads in flight together) and stride over the
// segment with the other blocks of that strip.  One entry per thread for height maps (a handful of
// blocks per frame: the kernel is one chain of round trips), four for value maps of many channels
// (the chip holds 2 K blocks at a time: with one entry per thread 40 channels were twenty rounds
// of that chain, 200 us; with four, 80 us).
constexpr int kCombineSlots = 32;       // (the lists hold about as many dead entries as live ones: twice the blocks of round 2)

// The combine kernel's share of the fill duty (StripArgs::defer_outer): the map rows z of (frame b,
// channel chl) outside the union window's rows [uz0, uz0 + uh) whose index among their wave's rows
// in the scatter kernel, j = z / (16 P), has (j & 7) >= head_share.  Block x of the frame's
// blocks takes the rows x, x + nblocks, ...; a row is stored by the block's threads as float4
// groups + their four mask bytes.  Issued in front of the chain of round trips the combine step
// is: the stores drain while it waits.
__device__ inline void combine_fill_outer(const StripCombineArgs& a, int b, int chl, int uz0, int uh) {
  const size_t fo = ((size_t)b * a.oc_total + a.ch0 + chl) * (size_t)a.mh * a.mw;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const f32x4 fv = {a.fill, a.fill, a.fill, a.fill};
  const int groups = a.mw >> 2;
  // (rows in flight per step: 256 threads cover 1024 cells)
  const int rows_per_step = groups >= kCombineThreads ? 1 : kCombineThreads / groups;
  const int sub = (int)threadIdx.x / groups, g0 = (int)threadIdx.x - sub * groups;
  const int stride = (int)gridDim.x * rows_per_step;
  for (int z0 = (int)blockIdx.x * rows_per_step; z0 < a.mh; z0 += stride) {
    const int z = z0 + (rows_per_step > 1 ? sub : 0);
    if (rows_per_step > 1 && sub >= rows_per_step) continue;
    if (z >= a.mh || (unsigned)(z - uz0) < (unsigned)uh) continue;
    if (((z / (16 * a.P)) & 7) < a.head_share) continue;
    float* const row = a.out + fo + (size_t)z * a.mw;
    uint8_t* const mrow = a.mask + fo + (size_t)z * a.mw;
    for (int g = rows_per_step > 1 ? g0 : (int)threadIdx.x; g < groups; g += kCombineThreads) {
      __builtin_nontemporal_store(fv, reinterpret_cast<f32x4*>(row) + g);
      reinterpret_cast<uint32_t*>(mrow)[g] = 0u;
    }
  }
}

// The groups of a frame's covers that no strip owns (the strips' shared-group lists): max / min
// over the slabs of the strips whose covers hold the group, written to the map with its mask
// bytes.  What a block needs of the frame (count, union window, the strips' windows) is
// wave-uniform.  One list entry per thread (height maps).
template <int RED>
__global__ void __launch_bounds__(kCombineThreads)
k_strip_combine_one(StripCombineArgs a) {
  const int fcl = blockIdx.y;                  // (frame of the launch) * oc + channel of the group
  const int bl = fcl / a.oc, b = a.b0 + bl;
  const int chl = fcl - bl * a.oc;
  const int nblocks = (int)gridDim.x;
  const int seg = (int)blockIdx.x % a.P, lane_block = (int)blockIdx.x / a.P;
  const int seg_blocks = (nblocks - seg + a.P - 1) / a.P;      // blocks that share this segment
  const int first = lane_block * kCombineThreads + (int)threadIdx.x;
  const uint32_t* const list = a.g_list + ((size_t)b * a.P + seg) * a.seg_cap;
  // (the thread's first entry is requested before the segment's length is known: one round trip
  // less in a kernel that is nothing but a chain of them)
  uint32_t entry = first < a.seg_cap ? list[first] : 0u;
  const int2 u_raw = *reinterpret_cast<const int2*>(a.g_unions + b);
  const int listed = min(a.g_counts[(size_t)b * strip::kMaxStrips + seg], a.seg_cap);
  const int ux0 = (short)(u_raw.x & 0xffff), uz0 = (short)(u_raw.x >> 16);
  // (the strips' windows: requested with the entry, in front of the early exit -- one round trip less)
  int2 wq[strip::kMaxStrips];
#pragma unroll
  for (int q = 0; q < strip::kMaxStrips; ++q)
    wq[q] = *reinterpret_cast<const int2*>(a.g_wins + (size_t)b * strip::kMaxStrips + (q < a.P ? q : 0));
  asm volatile("" : "+v"(wq[0].x), "+v"(entry));
  if (a.defer_outer)      // this kernel's share of the map rows outside the union window: fill value, mask 0
    combine_fill_outer(a, b, chl, uz0, (short)(u_raw.y >> 16));
  if (lane_block * kCombineThreads >= listed) return;
  const float* const slabs = a.slabs + ((size_t)(b * a.oc + chl) * a.P) * a.slab_stride;
  const size_t fo = ((size_t)b * a.oc_total + a.ch0 + chl) * (size_t)a.mh * a.mw;
  const float ident = RED == kMax ? -INFINITY : INFINITY;
  for (int i = first; i < listed; i += seg_blocks * kCombineThreads) {
    if (i != first) entry = list[i];
    if ((entry >> 24) == 0u) continue;       // (listed by a higher strip too: the lowest one's entry is the live one)
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
  const int fcl = blockIdx.y;                  // (frame of the launch) * oc + channel of the group
  const int bl = fcl / a.oc, b = a.b0 + bl;
  const int chl = fcl - bl * a.oc;
  const int nblocks = (int)gridDim.x;
  const int seg = (int)blockIdx.x % a.P, lane_block = (int)blockIdx.x / a.P;
  const int seg_blocks = (nblocks - seg + a.P - 1) / a.P;
  const int kStride = seg_blocks * kCombineThreads;
  const int first = lane_block * kCombineThreads + (int)threadIdx.x;
  const uint32_t* const list = a.g_list + ((size_t)b * a.P + seg) * a.seg_cap;
  // (the thread's first entries are requested before the segment's length is known: one round trip
  // less in a kernel that is nothing but a chain of them)
  uint32_t ent[kCombineEntries];
#pragma unroll
  for (int k = 0; k < kCombineEntries; ++k) ent[k] = first + k * kStride < a.seg_cap ? list[first + k * kStride] : 0u;
  const int listed = min(a.g_counts[(size_t)b * strip::kMaxStrips + seg], a.seg_cap);
  if (lane_block * kCombineThreads >= listed) return;
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
      if (base + k * kStride >= listed || (ent[k] >> 24) == 0u) continue;       // (past the end / a dead entry)
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

// Compact planes: the frame's shared groups by number -- where each lies in the map and which strips' covers hold
// it (meta), the strips' values of it at the same number in their planes: ONE round of coalesced loads per
// group, then max / min with the fill value and the map + mask stores.  Block x of a (frame, channel) takes the
// numbers x * 256 + thread, + gridDim.x * 256, ...; the first `spec_blocks` blocks request their first round
// before they know the frame's count (the numbers below the usual count are nearly always live).
struct PlaneCombineArgs {
  int b0, oc, ch0, oc_total, mh, mw;
  int P, plane_cap, spec_blocks;
  float fill;
  const int* g_counts;        // (B, kMaxStrips): [b][0] = shared groups of frame b
  const uint32_t* g_meta;     // (B, plane_cap)
  const float* planes;        // (B * oc, P, plane_cap) float4
  float* out;
  uint8_t* mask;
};

template <int RED>
__global__ void __launch_bounds__(kCombineThreads)
k_strip_combine_planes(PlaneCombineArgs a) {
  const int fcl = blockIdx.y;                  // (frame of the launch) * oc + channel of the group
  const int bl = fcl / a.oc, b = a.b0 + bl;
  const int chl = fcl - bl * a.oc;
  const int cap = a.plane_cap;
  const uint32_t* const meta = a.g_meta + (size_t)b * cap;
  const float4* const pl = reinterpret_cast<const float4*>(a.planes) + ((size_t)(b * a.oc + chl) * a.P) * (size_t)cap;
  const size_t fo = ((size_t)b * a.oc_total + a.ch0 + chl) * (size_t)a.mh * a.mw;
  const int stride = (int)gridDim.x * kCombineThreads;
  int i = (int)blockIdx.x * kCombineThreads + (int)threadIdx.x;       // (< plane_cap: the host sizes the grid)
  // (a vector load of the count: it returns in order with the round's other loads)
  const int n = min(*reinterpret_cast<const volatile int*>(a.g_counts + (size_t)b * strip::kMaxStrips), cap);
  if ((int)blockIdx.x >= a.spec_blocks && (int)blockIdx.x * kCombineThreads >= n) return;
  for (;;) {
    const uint32_t m = meta[i];
    float4 t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) t[q] = pl[(size_t)(q < a.P ? q : 0) * cap + i];
    if (i >= n) return;
    // (the fill value takes part: utils.py:470-477 reduces INTO the filled canvas)
    float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if ((m >> (28 + q)) & 1u) {
        acc.x = combine<RED>(acc.x, t[q].x); acc.y = combine<RED>(acc.y, t[q].y);
        acc.z = combine<RED>(acc.z, t[q].z); acc.w = combine<RED>(acc.w, t[q].w);
      }
    }
    const size_t cell = fo + ((size_t)(m & 0x0fffffffu) << 2);
    *reinterpret_cast<float4*>(a.out + cell) = acc;
    *reinterpret_cast<uint32_t*>(a.mask + cell) =
        (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
        ((uint32_t)mask_of(acc.z, a.fill) << 16) | ((uint32_t)mask_of(acc.w, a.fill) << 24);
    i += stride;
    if (i >= n) return;
  }
}

// (B, 32) dm_frame records in device memory -> StripPose records (test hook's input)
__global__ void __launch_bounds__(64)
k_pose_records(const float* frames, float* poses, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const float* f = frames + (size_t)b * 32;
  float* q = poses + (size_t)b * kPoseFloats;
  q[0] = f[10]; q[1] = f[12]; q[2] = f[16]; q[3] = f[18]; q[4] = f[19]; q[5] = f[20]; q[6] = f[21]; q[7] = f[22];
  q[8] = f[9]; q[9] = q[10] = q[11] = 0.0f;
}

// Test hook kernel: the geometry of every frame exactly as k_strip_scatter derives it (the same
// device function, the rig in the kernel arguments, the poses in device memory).
__global__ void __launch_bounds__(64)
k_strip_geometry_dump(strip::RigArgs rig, const float* poses, strip::FrameGeom* out) {
  __shared__ strip::FrameGeom g;
  const float* f = poses + (size_t)blockIdx.x * kPoseFloats;
  StripPose ps;
  ps.y0 = f[0]; ps.y2 = f[1]; ps.y6 = f[2]; ps.y8 = f[3]; ps.tx = f[4]; ps.tz = f[5]; ps.wo = f[6]; ps.ho = f[7];
  ps.cam_h = f[8]; ps.pad0 = ps.pad1 = ps.pad2 = 0.0f;
  __shared__ GeomLds gl;
  const int lane = (int)threadIdx.x;
  frame_geometry_wave64(rig, ps, lane, &gl);
  __syncthreads();
  if (lane < strip::kMaxStrips) { g.win[lane] = gl.win[lane]; g.L[lane] = gl.L[lane]; g.R[lane] = gl.R[lane]; }
  if (lane == 0) { g.U = gl.U; g.ok = gl.ok; g.inside = gl.inside; }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = g;
}

}  // namespace
}  // namespace dm
