// Device side of the LDS-windowed path: k_window_scatter, k_window_merge(_tiled),
// k_fuse_unions, k_fuse_windows and their argument structs.  Included by dm_window.hip only
// (the kernels are templates launched from that translation unit).
#pragma once

#include "dm_window_geometry.hpp"

namespace dm {
namespace {

// ---------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------
// Scalars of the scatter kernel, slimmed down to what it reads (SGPR budget).
// Disabled tests are encoded as always-true bounds (dmin = -inf, dmax = hmax =
// +inf: they then only reject NaN, which never reaches the map anyway) and a
// local-space projection as an identity yaw with zero translation, so the
// pixel loop has no flag to branch on.
// -DDM_STAMPS: instrumented build for tools/phase_stamps.py -- thread 0 of every
// workgroup of k_window_scatter records the 100 MHz real-time counter at phase boundaries.
#ifdef DM_STAMPS
#define DM_STAMP(k) do { long long t_; \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    stamp[k] = t_; } while (0)
#define DM_STAMPS_OUT() do { if (threadIdx.x == 0 && a.stamps) \
    for (int k_ = 0; k_ < 12; ++k_) \
      a.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 12 + k_] = stamp[k_]; \
  } while (0)
static long long* g_stamp_buffer = nullptr;
#else
#define DM_STAMP(k) do { } while (0)
#define DM_STAMPS_OUT() do { } while (0)
#endif

struct ScatterArgs {
  int W, H;
  int clip;                   // border pixels to drop (0 = none)
  int flip_h;
  float cx, cy, fx, fy, res;
  float fx_inv, fy_inv, res_inv;
  float dmin, dmax, hmax;
  float Hm1, mhm1;
  Parts parts;
  int dc, valid_c;
  int oc;                     // output channels per frame handled by this pass
  int ch0;                    // first output channel of this launch (channel groups)
  int oc_total;               // channels of `out` / `value`
  int slab_stride;            // cells per slab
  float fill;
#ifdef DM_STAMPS
  long long* stamps;
#endif
  int b0;                     // first frame of this launch's chunk
  Win16* g_wins;              // (B, nparts)  device copies for the kernels that follow
  Win16* g_unions;            // (B)
  const float* depth;
  const float* value;         // (B, oc_total, H, W) or NULL: project the heights
  const uint8_t* valid;
  float* slabs;
  // fill duty: the part of every output map outside its frame's union window
  float* out;
  uint8_t* mask;
  int mh, mw;
  // HAS_FLOW: the ego-motion flow grid of the same depth maps (camera_affine_grid, maps.py:353-460)
  // computed from the depth the projection has loaded anyway: (B) records in device memory, the
  // grid (B, dc, H, W, 2)
  const struct FlowRec* flow;
  float* flow_grid;
};

// What camera_affine_grid needs of a frame beyond the projection's own record (whose pitch
// rotation and camera height it shares): the rotation of the pose transition, its translation,
// and rotate([1,0,0], -cam_pitch).
struct FlowRec { float ry[9], tx, tz, ri[9], pad[4]; };
static_assert(sizeof(FlowRec) == 96, "FlowRec layout");

// (int)floorf(x) in one instruction; NaN -> 0, saturating (like v_cvt_i32_f32)
__device__ inline int floor_to_int(float x) {
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// global load and store of the wave (s_waitcnt vmcnt(0)), which would expose the latency
// of the depth rows and fill stores deliberately left in flight across it.
__device__ inline void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Reductions of the scatter and merge kernels.  kMin / kMax have the values of false / true
// (the batch-fuse kernels, max / min only, keep a bool).  kSum: ds_add_f32 into windows that
// start at 0, the fill value added once by the merge -- torch_scatter's "reduce into out";
// order dependent in float32, like torch_scatter's own GPU path (tolerance 1e-5).
// kMean: a sum window followed by a count window (uint32, ds_add_u32) in LDS and in the slab;
// the merge divides (fill + sum) by max(count, 1) -- torch_scatter's scatter_mean into `out`.
constexpr int kMin = 0, kMax = 1, kSum = 2, kMean = 3;
__host__ __device__ constexpr bool additive(int red) { return red == kSum || red == kMean; }

// ds_max_f32 / ds_min_f32: "store if new > old" -- torch_scatter's rule; a NaN
// operand never replaces a number.
template <int RED, class P>
__device__ inline void lds_reduce(P cell, float v) {
  if (RED == kMax) __hip_atomic_fetch_max(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else if (RED == kMin) __hip_atomic_fetch_min(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_add(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int RED>
__device__ inline float combine(float a, float b) {
  return RED == kMax ? fmaxf(a, b) : RED == kMin ? fminf(a, b) : a + b;
}

// FAST: every frame's rotations have the exact 0/1 pattern of rotate([1,0,0],.)
//       and rotate([0,1,0],.) AND the Markstein reciprocals are usable
//       (dm_pixel.hpp).  !FAST: full FMA chains and IEEE division.
// VEC = 4: 16-byte depth loads (W % 4 == 0, 16-byte aligned base); VEC = 1: any shape.
// HAS_VALUE: scatter value[b, ch] (maps.py:314-316) instead of the height.
// LEAN: finite depth bounds on both sides, no height truncation, no poisoned rays
//       (no border clip, no valid map): a non-finite or out-of-range pixel is then
//       already rejected by the two depth compares, so the ordered-compare and the
//       height compare are dropped.  (Pipeline-tail rows are poisoned through z.)
// HAS_FLOW (FAST, VEC = 4, LEAN, heights, no depth bands): every pixel's ego-motion flow (flow_pixel2,
//       dm_pixel.hpp) goes to a.flow_grid beside the projection -- one depth read for both.
// STREAM (lean variants): the depth maps are loaded non-temporally -- read once, they would otherwise evict
//       what the merge kernel reads back (the slabs); the host picks it for calls whose bytes cannot stay
//       cache resident and that no second reader of the depth maps follows (dm_window.hip).
template <int RED, bool FAST, bool HAS_VALID, bool HAS_VALUE, int VEC, bool LEAN, bool HAS_FLOW = false, bool STREAM = false>
__global__ void __launch_bounds__(kScatterThreads)
k_window_scatter(ScatterArgs a, const ScatterTables* __restrict__ tables) {
  static_assert(!STREAM || (LEAN && VEC == 4 && !HAS_FLOW), "the streaming variant exists for the lean kernels");
  static_assert(!HAS_FLOW || (FAST && VEC == 4 && LEAN && !HAS_VALUE && !HAS_VALID && !additive(RED)),
                "the fused flow rides on the lean height projection");
  const ScatterTables& t = *tables;            // this launch's chunk of frames
  extern __shared__ float lds[];
  const int part = blockIdx.x;                 // pr-major, pc-minor
  const int chl = blockIdx.y;                  // channel within this launch's group
  const int bl = blockIdx.z, b = a.b0 + bl;    // frame within the chunk / in the batch
  const int ch = a.ch0 + chl;                  // output channel
  const int dch = a.dc == 1 ? 0 : ch;          // depth / cell-index channel (utils.py:475-477)
  const int nparts = a.parts.pc * a.parts.pr * a.parts.pd;
  const int pcx = part % a.parts.pc, pry = (part / a.parts.pc) % a.parts.pr;
  const int pdk = part / (a.parts.pc * a.parts.pr);          // depth band
  // The part's pixel rectangle and this thread's place in it need nothing but kernel
  // arguments, so the first depth rows are requested before anything of the staged
  // table has arrived (window, union window, frame record: all of the head of the kernel
  // runs under these loads).
  const int q0 = pcx * a.parts.wp;
  int q1 = q0 + a.parts.wp; if (q1 > a.W) q1 = a.W;
  const int r0 = pry * a.parts.hp;
  int r1 = r0 + a.parts.hp; if (r1 > a.H) r1 = a.H;
  const int nx = (q1 - q0 + VEC - 1) / VEC;    // lane groups per row
  const int ntx = nx < kScatterThreads ? nx : kScatterThreads;
  const int rows_per_iter = kScatterThreads / ntx;
  const int gx = threadIdx.x % ntx, gy = threadIdx.x / ntx;
  const size_t N = (size_t)a.H * a.W;
  const float* dimg = a.depth + ((size_t)b * a.dc + dch) * N;
  const uint8_t* vimg = HAS_VALID
      ? a.valid + ((size_t)b * a.valid_c + (a.valid_c == 1 ? 0 : dch)) * N : nullptr;
  const float* simg = HAS_VALUE ? a.value + ((size_t)b * a.oc_total + ch) * N : nullptr;
  const float qnan = __builtin_nanf("");
  // rows of a thread in flight per pipeline stage: value maps carry a second float4 per row and spilled 21-95
  // vector registers into scratch at four (the kernel is capped at 128 VGPRs by its 1024 threads;
  // tests/test_codegen_guard.py) -- two, as in k_strip_scatter
  constexpr int kRowsInFlight = (HAS_VALUE && VEC == 4) ? 2 : dm::kRowsInFlight;
  float za[kRowsInFlight][VEC], zb_[kRowsInFlight][VEC];
  float va[HAS_VALUE ? kRowsInFlight : 1][VEC], vb_[HAS_VALUE ? kRowsInFlight : 1][VEC];
  auto load_rows_at = [&](float (&z)[kRowsInFlight][VEC],
                          float (&sv)[HAS_VALUE ? kRowsInFlight : 1][VEC], int q, int r) {
#pragma unroll
    for (int u = 0; u < kRowsInFlight; ++u) {
      // rows past the part are clamped to its last row (a legal address);
      // project_rows() ignores them.  No per-lane branch: the loads stay in
      // one basic block and the compiler can wait on them individually.
      int rr = r + u * rows_per_iter;
      rr = rr < r1 ? rr : r1 - 1;
      if (VEC == 4) {
        typedef float wf32x4 __attribute__((ext_vector_type(4)));
        const wf32x4* const src4 = reinterpret_cast<const wf32x4*>(dimg + (size_t)rr * a.W + q);
        const wf32x4 t = STREAM ? __builtin_nontemporal_load(src4) : *src4;
        z[u][0] = t.x; z[u][1 % VEC] = t.y; z[u][2 % VEC] = t.z; z[u][3 % VEC] = t.w;
      } else {
        z[u][0] = dimg[(size_t)rr * a.W + q];
      }
      if (HAS_VALID) {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
          z[u][k] = vimg[(size_t)rr * a.W + q + k] ? z[u][k] : qnan;
      }
      if (HAS_VALUE) {
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(simg + (size_t)rr * a.W + q);
          sv[u][0] = t.x; sv[u][1 % VEC] = t.y; sv[u][2 % VEC] = t.z; sv[u][3 % VEC] = t.w;
        } else {
          sv[u][0] = simg[(size_t)rr * a.W + q];
        }
      }
    }
  };
  bool first_rows_loaded = false;
  if (gx < nx && q1 > q0 && r1 > r0) {         // (always, for a non-empty part)
    load_rows_at(za, va, q0 + gx * VEC, r0 + gy);
    first_rows_loaded = true;
  }

  // Everything this workgroup reads from the staged table -- its window, the frame's union
  // window and the frame record -- is requested in ONE batch of scalar loads and pinned
  // (the asm makes the values opaque): left to the compiler these loads trickle in close
  // to their first use, one dependent ~0.7-us round trip after the other.
  int w_raw[2], u_raw[2];
  float fr[23];
  {
    const int* tw = reinterpret_cast<const int*>(&t.wins[bl * kFewParts + (part & (kFewParts - 1))]);
    const int* tu = reinterpret_cast<const int*>(&t.unions[bl]);
    const float* tf = reinterpret_cast<const float*>(&t.frames[bl]);
    w_raw[0] = tw[0]; w_raw[1] = tw[1]; u_raw[0] = tu[0]; u_raw[1] = tu[1];
#pragma unroll
    for (int i = 0; i < 23; ++i) fr[i] = tf[i];
    asm volatile("" : "+s"(w_raw[0]), "+s"(w_raw[1]), "+s"(u_raw[0]), "+s"(u_raw[1]),
                      "+s"(fr[0]), "+s"(fr[1]), "+s"(fr[2]), "+s"(fr[3]), "+s"(fr[4]), "+s"(fr[5]),
                      "+s"(fr[6]), "+s"(fr[7]), "+s"(fr[8]), "+s"(fr[9]), "+s"(fr[10]), "+s"(fr[11]),
                      "+s"(fr[12]), "+s"(fr[13]), "+s"(fr[14]), "+s"(fr[15]), "+s"(fr[16]),
                      "+s"(fr[17]), "+s"(fr[18]), "+s"(fr[19]), "+s"(fr[20]), "+s"(fr[21]),
                      "+s"(fr[22]));
  }
  // (HAS_FLOW: the frame's flow record, requested with the rest)
  float fl_ry[9], fl_ri[9], fl_tx = 0.0f, fl_tz = 0.0f;
  if (HAS_FLOW) {
    // (through the constant address space and pinned: the kernel's stores to the flow grid would
    // otherwise turn these wave-uniform loads into vector loads held in twenty VGPRs)
    typedef const __attribute__((address_space(4))) float cfloat;
    const uintptr_t fp = reinterpret_cast<uintptr_t>(a.flow + b);
    cfloat* tf = (cfloat*)(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(fp >> 32)) << 32) |
                           (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(fp & 0xffffffffu)));
#pragma unroll
    for (int i = 0; i < 9; ++i) { fl_ry[i] = tf[i]; fl_ri[i] = tf[11 + i]; }
    fl_tx = tf[9]; fl_tz = tf[10];
    asm volatile("" : "+s"(fl_ry[0]), "+s"(fl_ry[1]), "+s"(fl_ry[2]), "+s"(fl_ry[3]), "+s"(fl_ry[4]),
                      "+s"(fl_ry[5]), "+s"(fl_ry[6]), "+s"(fl_ry[7]), "+s"(fl_ry[8]), "+s"(fl_tx), "+s"(fl_tz),
                      "+s"(fl_ri[0]), "+s"(fl_ri[1]), "+s"(fl_ri[2]), "+s"(fl_ri[3]), "+s"(fl_ri[4]),
                      "+s"(fl_ri[5]), "+s"(fl_ri[6]), "+s"(fl_ri[7]), "+s"(fl_ri[8]));
  }
  const Win16 w_few = {(short)(w_raw[0] & 0xffff), (short)(w_raw[0] >> 16),
                       (short)(w_raw[1] & 0xffff), (short)(w_raw[1] >> 16)};
  const Window w = nparts <= kFewParts ? widen(w_few) : widen(t.wins[bl * nparts + part]);
  const int area = w.w * w.h;                  // 0: nothing of this part can land
#ifdef DM_STAMPS
  long long stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  DM_STAMP(0);
  // younger waves of a SIMD get the higher issue priority (age arbitration favours the
  // oldest wave otherwise, and the last wave left on a SIMD runs latency bound)
  {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave >= 12) __builtin_amdgcn_s_setprio(3);
    else if (wave >= 8) __builtin_amdgcn_s_setprio(2);
    else if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  }

  // Fill duty, interleaved with the scatter so that these stores ride under the
  // projection: map rows part, part + nparts, ... of (b, ch), minus the frame's union
  // window U (k_window_merge writes U).  Wave-level, as in k_strip_scatter: wave v takes the
  // rows part + (v + 16 j) nparts, one step stores 256 cells of a row (float4 per lane) and their
  // mask bytes with SCALAR addressing -- the map of this (frame, channel) as a raw buffer
  // resource, the row and chunk in the scalar offset, the lane's fixed 16 / 4 bytes in the vector
  // offset.  A lane with nothing to write (inside U, past the row's end, past the wave's rows)
  // gets a vector offset past the end of the buffer and is dropped by the hardware's range
  // check: unconditional straight-line stores (the compiler can count the pipelined loop's
  // waits exactly) at a handful of VALU instructions per KB.
  const Window U = {(short)(u_raw[0] & 0xffff), (short)(u_raw[0] >> 16),
                    (short)(u_raw[1] & 0xffff), (short)(u_raw[1] >> 16)};
  const int fill_rows = (a.mh - part + nparts - 1) / nparts;
  const size_t map_base = ((size_t)b * a.oc_total + ch) * (size_t)a.mh * a.mw;
  const bool do_fill = a.out != nullptr && fill_rows > 0;                       // wave-uniform
  const int f_wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int f_lane4 = ((int)threadIdx.x & 63) << 2;
  const int f_chunks = (a.mw + 255) >> 8;
  const int fill_steps = do_fill && f_wave < fill_rows ? ((fill_rows - f_wave + 15) >> 4) * f_chunks : 0;
  typedef unsigned int fill_u32x4 __attribute__((ext_vector_type(4)));
  const unsigned f_cells = (unsigned)a.mh * (unsigned)a.mw;     // (< 2^28: window_path_supported)
  const __amdgpu_buffer_rsrc_t rs_fill_out =
      __builtin_amdgcn_make_buffer_rsrc(a.out + map_base, 0, do_fill ? f_cells * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_fill_mask =
      __builtin_amdgcn_make_buffer_rsrc(a.mask + map_base, 0, do_fill ? f_cells : 0u, 0x00020000);
  const unsigned f_bits = __float_as_uint(a.fill);
  int fs = 0, f_row = part + f_wave * nparts, f_chunk = 0;                      // (wave-uniform)
  auto fill_step = [&]() {
    const int x = (f_chunk << 8) + f_lane4;
    const bool live = fs < fill_steps;                                          // (scalar)
    const bool in_rows = (unsigned)(f_row - U.z0) < (unsigned)U.h;              // (scalar)
    const bool skip = !live | (x >= a.mw) | (in_rows & ((unsigned)(x - U.x0) < (unsigned)U.w));
    // (the scalar offset must be the same in every lane, skipping or not)
    const int cell0 = __builtin_amdgcn_readfirstlane(live ? f_row * a.mw + (f_chunk << 8) : 0);
    buffer_store_b128_at_scalar_offset((fill_u32x4){f_bits, f_bits, f_bits, f_bits}, rs_fill_out,
                                       skip ? 0x7ffffff0 : f_lane4 << 2, cell0 << 2);
    __builtin_amdgcn_raw_buffer_store_b32(0u, rs_fill_mask, skip ? 0x7ffffff0 : f_lane4, cell0, 0);
    ++fs;
    const bool next_row = f_chunk + 1 == f_chunks;
    f_chunk = next_row ? 0 : f_chunk + 1;
    f_row += next_row ? 16 * nparts : 0;
  };
  // device copies of the geometry for the kernels that follow (stored at the very end: a
  // store in flight makes its wave wait before the first depth loads)
  auto publish_geometry = [&]() {
    if (chl == 0 && threadIdx.x == 0) {
      a.g_wins[(size_t)b * nparts + part] = narrow16(w);
      if (part == 0) a.g_unions[b] = narrow16(U);
    }
  };
  if (area == 0) {                             // wave-uniform
    while (fs < fill_steps) fill_step();
    publish_geometry();
    return;
  }
  // FrameRec: p[9], cam_h, y[9], tx, tz, wo, ho
  // pitch: rows 1,2 of R; yaw: rows 0,2 (the rest is 0/1 when FAST)
  const float p0 = fr[0], p1 = fr[1], p2 = fr[2], p3 = fr[3], p4 = fr[4],
              p5 = fr[5], p6 = fr[6], p7 = fr[7], p8 = fr[8];
  const float y0 = fr[10], y1r = fr[11], y2r = fr[12], y3 = fr[13], y4 = fr[14],
              y5 = fr[15], y6 = fr[16], y7 = fr[17], y8 = fr[18];
  const float cam_h = fr[9], tx = fr[19], tz = fr[20];
  const float wo = fr[21], ho = fr[22];
  const float flip_s = a.flip_h ? -1.0f : 1.0f, flip_c = a.flip_h ? a.mhm1 : 0.0f;
  FlowPairs flow_pairs;      // (HAS_FLOW: the flow's coefficients as {c, c} pairs, flow_pixel2)
  if (HAS_FLOW) {
    auto pair = [](float x) { return (f32x2){x, x}; };
    const float rp9[9] = {p0, p1, p2, p3, p4, p5, p6, p7, p8};
#pragma unroll
    for (int i = 0; i < 9; ++i) { flow_pairs.rp[i] = pair(rp9[i]); flow_pairs.ry[i] = pair(fl_ry[i]); flow_pairs.ri[i] = pair(fl_ri[i]); }
    flow_pairs.cam_h = pair(cam_h); flow_pairs.neg_cam_h = pair(-cam_h); flow_pairs.tx = pair(fl_tx); flow_pairs.tz = pair(fl_tz);
    flow_pairs.fx = pair(a.fx); flow_pairs.cx = pair(a.cx); flow_pairs.fy = pair(a.fy); flow_pairs.cy = pair(a.cy);
    flow_pairs.zero = pair(0.0f); flow_pairs.eps = pair(1e-7f); flow_pairs.Hm1 = pair(a.Hm1);
  }
  DM_STAMP(1);
  bool lds_ready = false;
  const unsigned dummy = (unsigned)(RED == kMean ? 2 * area : area) + (threadIdx.x & 63u);   // 64 scratch cells after the window(s)
  float band_lo = a.dmin, band_hi = a.dmax;    // this part's depth band (wave-uniform)
  if (a.parts.pd > 1) band_bounds(a.dmin, a.dmax, a.parts.pd, pdk, band_lo, band_hi, a.parts.geo);
  // neighbouring bands share their boundary value: harmless for max / min, counted twice by
  // a sum, which therefore takes every band but the last half open
  const bool last_band = pdk == a.parts.pd - 1;
  const float lds_init = additive(RED) ? 0.0f : a.fill;
  const int cells = RED == kMean ? 2 * area : area;   // mean: sum window, then count window

  {
    for (int g = gx; g < nx; g += ntx) {       // one trip unless the strip is wider than the block
      const int q = q0 + g * VEC;
      // ray slope of each column (maps.py:677); border columns are poisoned with
      // NaN, which flows through X to the cell coordinates (maps.py:48-70)
      float ax[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        const float d = (float)(q + k) - a.cx;
        ax[k] = FAST ? div_markstein(d, a.fx, a.fx_inv) : d / a.fx;
        // (idle threads and pipeline-tail rows simply repeat the part's last row: a max /
        // min reduction is idempotent, so they need no poison)
        if (!LEAN) ax[k] = (q + k < a.clip || q + k >= a.W - a.clip) ? qnan : ax[k];
      }
      // Software pipeline over groups of kRowsInFlight rows: the loads of group
      // i+1 are in flight while group i is projected (all waves of a workgroup
      // run in phase, so latency has to be hidden inside each wave).
      const int step = rows_per_iter * kRowsInFlight;
      auto load_rows = [&](float (&z)[kRowsInFlight][VEC],
                           float (&sv)[HAS_VALUE ? kRowsInFlight : 1][VEC], int r) {
        load_rows_at(z, sv, q, r);
      };
      // (always_inline: called twice per pipeline trip; past a size -- the fused flow -- the inliner
      // would leave it a function of its own, with every capture handed over through scratch)
      auto project_rows = [&](const float (&z)[kRowsInFlight][VEC],
                              const float (&sv)[HAS_VALUE ? kRowsInFlight : 1][VEC], int r)
                              __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
          int rr = r + u * rows_per_iter;
          // (a sum counts every pixel once: rows past the part and the idle threads of a
          // block whose width does not divide 1024 must not repeat a row)
          const bool own_row = !additive(RED) || (rr < r1 && gy < rows_per_iter);
          rr = rr < r1 ? rr : r1 - 1;                            // tail: repeat the last row
          float yr = (float)rr;
          yr = a.flip_h ? a.Hm1 - yr : yr;                       // maps.py:670-671
          const float dy = yr - a.cy;
          float ay = FAST ? div_markstein(dy, a.fy, a.fy_inv) : dy / a.fy;
          // rows in the clipped border: poison
          if (!LEAN) ay = (rr < a.clip || rr >= a.H - a.clip) ? qnan : ay;
          // the VEC pixels of a row are projected side by side (independent
          // chains for the scheduler); their LDS atomics come last so that no
          // branch separates the arithmetic of neighbouring pixels
          unsigned li[VEC];
          float hv[VEC];
          bool ok[VEC];
          float xfv[VEC], zfv[VEC], h1v[VEC], h2v[VEC];
          if (FAST && VEC == 4) {
            // two pixels per instruction (v_pk_mul/fma/add_f32): the kernel is bound by
            // dependent-instruction issue, and the packed forms have the same rounding
            // per element as the scalar ones
            typedef float f2 __attribute__((ext_vector_type(2)));
            if (HAS_FLOW) {
              // The flow of the row's four pixels FIRST, two pixels at a time, each pair stored (16
              // bytes, non-temporal) before anything else is computed -- scheduling barriers keep the
              // compiler from interleaving it with the projection (all of it at once needs ~50 more
              // registers than the kernel's 128).  The stand-alone kernel's arithmetic (flow_pixel2)
              // on the same X = ax * z, Y = ay * z.  (Tail rows and idle threads repeat a row: the
              // same values to the same place.)
              typedef float f32x4 __attribute__((ext_vector_type(4)));
              f32x4* dst = reinterpret_cast<f32x4*>(
                  a.flow_grid + 2 * (((size_t)b * a.dc + dch) * N + (size_t)rr * a.W + q));
#pragma unroll
              for (int k = 0; k < VEC; k += 2) {
                const f32x2 zz = {z[u][k], z[u][k + 1]};
                const f32x2 axp = {ax[k], ax[k + 1]};
                f32x2 fu, fw;
                flow_pixel2(zz, axp * zz, (f32x2){ay, ay} * zz, flow_pairs, a.flip_h != 0, fu, fw);
                __builtin_nontemporal_store((f32x4){fu.x, fw.x, fu.y, fw.y}, dst + (k >> 1));
                __builtin_amdgcn_sched_barrier(0);
              }
            }
#pragma unroll
            for (int k = 0; k < VEC; k += 2) {
              const f2 zz = {z[u][k], z[u][(k + 1) % VEC]};
              const f2 axp = {ax[k], ax[(k + 1) % VEC]};
              const f2 X = axp * zz;
              const f2 Y = zz * ay;                                            // maps.py:677-678
              f2 h1 = __builtin_elementwise_fma(zz, (f2){p7, p7}, Y * p4) + cam_h;   // maps.py:790-797
              const f2 z1 = __builtin_elementwise_fma(zz, (f2){p8, p8}, Y * p5);
              const f2 x2 = __builtin_elementwise_fma(z1, (f2){y6, y6}, X * y0) + tx;   // maps.py:884-892
              const f2 z2 = __builtin_elementwise_fma(z1, (f2){y8, y8}, X * y2r) + tz;
              const f2 ri = {a.res_inv, a.res_inv}, nres = {-a.res, -a.res};
              const f2 qx = x2 * ri, qz = z2 * ri;                             // exact division
              f2 xf2 = __builtin_elementwise_fma(__builtin_elementwise_fma(nres, qx, x2), ri, qx) + wo;
              f2 zf2 = __builtin_elementwise_fma(__builtin_elementwise_fma(nres, qz, z2), ri, qz) + ho;
              zf2 = __builtin_elementwise_fma(zf2, (f2){flip_s, flip_s}, (f2){flip_c, flip_c});
              xf2 = xf2 + 0.5f;
              zf2 = zf2 + 0.5f;
              xfv[k] = xf2.x; xfv[(k + 1) % VEC] = xf2.y;
              zfv[k] = zf2.x; zfv[(k + 1) % VEC] = zf2.y;
              h1v[k] = h1.x; h1v[(k + 1) % VEC] = h1.y;
              h2v[k] = h1.x; h2v[(k + 1) % VEC] = h1.y;
            }
          } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
              const float zz = z[u][k];
              const float X = ax[k] * zz, Y = ay * zz;             // maps.py:677-678
              float x1, h1, z1, x2, h2, z2;
              if (FAST) {
                x1 = X;
                h1 = __builtin_fmaf(zz, p7, Y * p4) + cam_h;       // maps.py:790-797
                z1 = __builtin_fmaf(zz, p8, Y * p5);
                x2 = __builtin_fmaf(z1, y6, x1 * y0) + tx;         // maps.py:884-892
                z2 = __builtin_fmaf(z1, y8, x1 * y2r) + tz;
                h2 = h1;
              } else {
                x1 = __builtin_fmaf(zz, p6, __builtin_fmaf(Y, p3, X * p0)) + 0.0f;
                h1 = __builtin_fmaf(zz, p7, __builtin_fmaf(Y, p4, X * p1)) + cam_h;
                z1 = __builtin_fmaf(zz, p8, __builtin_fmaf(Y, p5, X * p2)) + 0.0f;
                x2 = __builtin_fmaf(z1, y6, __builtin_fmaf(h1, y3, x1 * y0)) + tx;
                h2 = __builtin_fmaf(z1, y7, __builtin_fmaf(h1, y4, x1 * y1r)) + 0.0f;
                z2 = __builtin_fmaf(z1, y8, __builtin_fmaf(h1, y5, x1 * y2r)) + tz;
              }
              float xf = (FAST ? div_markstein(x2, a.res, a.res_inv) : x2 / a.res) + wo;
              float zf = (FAST ? div_markstein(z2, a.res, a.res_inv) : z2 / a.res) + ho;
              // flip: (mh-1) - zf as fma(zf, -1, mh-1); no flip: fma(zf, 1, 0) -- both exact
              zf = __builtin_fmaf(zf, flip_s, flip_c);             // maps.py:1006-1009
              xfv[k] = xf + 0.5f;                                  // maps.py:1012-1013
              zfv[k] = zf + 0.5f;
              h1v[k] = h1; h2v[k] = h2;
            }
          }
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            const float zz = z[u][k], xf = xfv[k], zf = zfv[k], h1 = h1v[k], h2 = h2v[k];
            // floor + convert in one instruction; window test in integers (the
            // window lies inside the map).  The conversion saturates and maps
            // NaN to 0, so NaN is excluded by the ordered compare.
            // maps.py:537-544, 286-288, 1150-1158
            const unsigned ux = (unsigned)(floor_to_int(xf) - w.x0);
            const unsigned uz = (unsigned)(floor_to_int(zf) - w.z0);
            ok[k] = ux < (unsigned)w.w && uz < (unsigned)w.h && zz >= band_lo &&
                    (additive(RED) && !last_band ? zz < band_hi : zz <= band_hi) && own_row;
            if (!LEAN) ok[k] = ok[k] && !__builtin_isunordered(xf, zf) && h1 <= a.hmax;
            if (!FAST && !HAS_VALUE) ok[k] = ok[k] && (h2 == h2);
            const float sval = HAS_VALUE ? sv[u][k] : h2;
            if (HAS_VALUE) ok[k] = ok[k] && (sval == sval);      // NaN never replaces a number
            // rejected pixels are redirected to a per-lane dummy cell behind the
            // window instead of being branched around: the whole row group stays
            // one basic block the scheduler can interleave
            unsigned cell = __umul24(uz, (unsigned)w.w) + ux;
            asm("" : "+v"(cell));   // keep the select below a v_cndmask, not a branch
            li[k] = ok[k] ? cell : dummy;
            hv[k] = sval;
          }
          // Neighbouring pixels of a row often share a cell (walls, near floor), and lanes
          // that hit one LDS address serialise: +30 % kernel time on scene-like depth.  If
          // any thread of the wave has its four pixels in ONE cell (wave-uniform test), every
          // run of equal cells inside a thread is reduced in registers and only its last
          // pixel issues the atomic.  (Unconditionally the extra selects cost 2.6 us on
          // incoherent depth; behind the test 1 us.)
          if (VEC == 4 && RED != kMean &&
              __builtin_amdgcn_ballot_w64(li[0] == li[VEC - 1] && li[0] != dummy) != 0) {
#pragma unroll
            for (int k = 0; k + 1 < VEC; ++k) {
              const bool same = li[k] == li[k + 1];
              const float m = combine<RED>(hv[k], hv[k + 1]);
              hv[k + 1] = same ? m : hv[k + 1];
              li[k] = same ? dummy : li[k];
            }
          }
#pragma unroll
          for (int k = 0; k < VEC; ++k) {
            // ds_add_f32 runs at 206 G/s on MI355X (ds_max_f32: 4 850 G/s, tools/microbench):
            // a sum does not pay for the lanes of rejected pixels, it masks them off
            if (!additive(RED) || li[k] != dummy) {
              lds_reduce<RED == kMean ? kSum : RED>(lds + li[k], hv[k]);
              if (RED == kMean)
                __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(lds + area) + li[k], 1u,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
      };
      const int niter = (r1 - r0 + step - 1) / step;             // wave-uniform
      // Two copies of the pipelined loop, with and without fill duty, chosen by ONE
      // wave-uniform branch: inside, exactly kFillPerHalf unconditional fill stores
      // follow each group of loads, so the waits count them and never wait on a store
      // or on the prefetch.
      auto pipeline = [&](auto with_fill) {
        constexpr bool kFill = decltype(with_fill)::value;
        int r = r0 + gy;
        if (!first_rows_loaded) load_rows(za, va, r);      // (later trips of the column loop)
        first_rows_loaded = false;
        DM_STAMP(2);
        // the LDS window is initialised while the first depth rows are in flight
        if (!lds_ready) {                      // wave-uniform, first trip only
          for (int i = threadIdx.x * 4; i < cells; i += kScatterThreads * 4)
            *reinterpret_cast<float4*>(lds + i) = make_float4(lds_init, lds_init, lds_init, lds_init);
          lds_barrier();
          lds_ready = true;
        }
        DM_STAMP(3);
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
  DM_STAMP(4);
  while (fs < fill_steps) fill_step();
  lds_barrier();
  DM_STAMP(5);
  const int pid = (b * a.oc + chl) * nparts + part;      // slabs are per channel group
  float* slab = a.slabs + (size_t)pid * a.slab_stride;
  for (int i = threadIdx.x * 4; i < cells; i += kScatterThreads * 4)
    *reinterpret_cast<float4*>(slab + i) = *reinterpret_cast<const float4*>(lds + i);
  publish_geometry();
  DM_STAMP(6);
  DM_STAMPS_OUT();
}

// scatter_mean into `out` (utils.py:470-477 with torch_scatter's out= semantics): the fill value
// takes part in the numerator, the count is clamped to 1.
__device__ inline void mean_of(float4& acc, uint4 cnt) {
  acc.x = acc.x / (float)(cnt.x < 1u ? 1u : cnt.x);
  acc.y = acc.y / (float)(cnt.y < 1u ? 1u : cnt.y);
  acc.z = acc.z / (float)(cnt.z < 1u ? 1u : cnt.z);
  acc.w = acc.w / (float)(cnt.w < 1u ? 1u : cnt.w);
}

struct MergeArgs {
  int b0, oc, ch0, oc_total, mh, mw;  // oc channels per frame in this launch, starting at ch0
  int nparts;                 // pc * pr
  int slab_stride;
  float fill;
  // this chunk's part windows (row stride win_stride) and union windows: in the staged
  // table (already in every XCD's L2)
  const Win16* wins;
  const Win16* unions;
  int win_stride;
  const float* slabs;
  float* out;
  uint8_t* mask;
};

constexpr int kMergeThreads = 256;
constexpr int kMergeDepth = 8;      // slab loads a thread of the merge kernels keeps in flight

// Writes the union window U of every (frame, channel): max/min over the slabs covering
// each cell, fill where none does.  One float4 group per thread, row-major inside U, so
// a wave's store covers up to 1 KiB (map) / 256 B (mask) contiguously.
template <int RED>
__global__ void __launch_bounds__(kMergeThreads)
k_window_merge(MergeArgs a) {
  const int fcl = blockIdx.y;                  // (frame in chunk) * oc + channel of the group
  const int bl = fcl / a.oc, b = a.b0 + bl;
  const int fc = fcl + a.b0 * a.oc;            // slab index of (frame, channel)
  const size_t fo = (size_t)b * a.oc_total + a.ch0 + (fcl - bl * a.oc);   // map index in `out`
  const Window U = widen(a.unions[bl]);
  const int win_stride = a.win_stride;
  const int ug4 = U.w >> 2;                    // float4 groups per U row
  const int total = ug4 * U.h;
  const int i = blockIdx.x * kMergeThreads + threadIdx.x;
  if (i >= total) return;
  const int row = i / ug4;
  const int zb = U.z0 + row, x = U.x0 + ((i - row * ug4) << 2);
  float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
  uint4 cnt = make_uint4(0u, 0u, 0u, 0u);      // mean: points per cell
  // kMergeDepth parts at a time: which of their windows hold the group first, then their slab loads all in
  // flight together, then the reduction in part order (part by part -- test, load, wait, combine -- a group under
  // six windows was six dependent round trips: 11.5 us per launch at cfg5)
  for (int p0 = 0; p0 < a.nparts; p0 += kMergeDepth) {
    float4 s[kMergeDepth];
    uint4 c[kMergeDepth];
    bool hit[kMergeDepth];
#pragma unroll
    for (int k = 0; k < kMergeDepth; ++k) {
      const int p = p0 + k < a.nparts ? p0 + k : a.nparts - 1;
      const Window w = widen(a.wins[bl * win_stride + p]);
      const unsigned ux = (unsigned)(x - w.x0), uz = (unsigned)(zb - w.z0);
      hit[k] = p0 + k < a.nparts && w.w != 0 && ux < (unsigned)w.w && uz < (unsigned)w.h;
      const float* slab = a.slabs + ((size_t)fc * a.nparts + p) * a.slab_stride;
      s[k] = acc;
      c[k] = cnt;
      if (hit[k]) {
        s[k] = *reinterpret_cast<const float4*>(slab + (size_t)uz * w.w + ux);
        if (RED == kMean)
          c[k] = *reinterpret_cast<const uint4*>(slab + (size_t)w.w * w.h + (size_t)uz * w.w + ux);
      }
    }
#pragma unroll
    for (int k = 0; k < kMergeDepth; ++k) {
      if (!hit[k]) continue;
      if (RED == kMean) { cnt.x += c[k].x; cnt.y += c[k].y; cnt.z += c[k].z; cnt.w += c[k].w; }
      acc.x = combine<RED>(acc.x, s[k].x);
      acc.y = combine<RED>(acc.y, s[k].y);
      acc.z = combine<RED>(acc.z, s[k].z);
      acc.w = combine<RED>(acc.w, s[k].w);
    }
  }
  if (RED == kMean) mean_of(acc, cnt);
  const size_t cell = fo * (size_t)a.mh * a.mw + (size_t)zb * a.mw + x;
  *reinterpret_cast<float4*>(a.out + cell) = acc;
  const uint32_t mk = (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
                      ((uint32_t)mask_of(acc.z, a.fill) << 16) |
                      ((uint32_t)mask_of(acc.w, a.fill) << 24);
  *reinterpret_cast<uint32_t*>(a.mask + cell) = mk;
}

// The same for frames of many parts (image parts x depth bands): a block owns a tile of
// kTileGroups x kTileRows float4 groups of U, first lists the parts whose windows touch the
// tile (in part order, so that the result does not depend on the tiling), and its threads
// then visit only those instead of all of them.
constexpr int kTileGroups = 16, kTileRows = 16;
constexpr int kTiledMergeParts = 16;      // frames of at least this many parts take the tiled merge
static_assert(kTileGroups * kTileRows == kMergeThreads && kMaxParts <= kMergeThreads, "one test per thread");

template <int RED>
__global__ void __launch_bounds__(kMergeThreads)
k_window_merge_tiled(MergeArgs a) {
  __shared__ Win16 lwin[kMaxParts];
  __shared__ short lpart[kMaxParts];
  __shared__ int wave_hits[kMergeThreads / 64];
  const int fcl = blockIdx.y;
  const int bl = fcl / a.oc, b = a.b0 + bl;
  const int fc = fcl + a.b0 * a.oc;
  const size_t fo = (size_t)b * a.oc_total + a.ch0 + (fcl - bl * a.oc);
  const Window U = widen(a.unions[bl]);
  const int win_stride = a.win_stride;
  const int tiles_x = ((U.w >> 2) + kTileGroups - 1) / kTileGroups;
  const int tiles_z = (U.h + kTileRows - 1) / kTileRows;
  const int t = blockIdx.x;
  if (t >= tiles_x * tiles_z) {                  // the whole block
    return;
  }
  const int tz = t / tiles_x, tx = t - tz * tiles_x;
  const int x0 = U.x0 + tx * (kTileGroups * 4), z0 = U.z0 + tz * kTileRows;
  const int x1 = min(x0 + kTileGroups * 4, U.x0 + U.w), z1 = min(z0 + kTileRows, U.z0 + U.h);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  Win16 mine = Win16{0, 0, 0, 0};
  bool hit = false;
  if (tid < a.nparts) {
    mine = a.wins[bl * win_stride + tid];
    hit = mine.w > 0 && mine.x0 < x1 && mine.x0 + mine.w > x0 && mine.z0 < z1 && mine.z0 + mine.h > z0;
  }
  const unsigned long long votes = __builtin_amdgcn_ballot_w64(hit);
  if (lane == 0) wave_hits[wave] = __builtin_popcountll(votes);
  __syncthreads();
  int before = 0, n = 0;
#pragma unroll
  for (int w = 0; w < kMergeThreads / 64; ++w) {
    before += w < wave ? wave_hits[w] : 0;
    n += wave_hits[w];
  }
  if (hit) {
    const int at = before + __builtin_popcountll(votes & ((1ull << lane) - 1ull));
    lwin[at] = mine;
    lpart[at] = (short)tid;
  }
  __syncthreads();

  const int x = x0 + ((tid & (kTileGroups - 1)) << 2), zb = z0 + tid / kTileGroups;
  if (x >= x1 || zb >= z1) {
    return;
  }
  float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
  uint4 cnt = make_uint4(0u, 0u, 0u, 0u);      // mean: points per cell
  const float* slabs = a.slabs + (size_t)fc * a.nparts * a.slab_stride;
  for (int j0 = 0; j0 < n; j0 += kMergeDepth) {        // (as in k_window_merge: the loads of kMergeDepth candidates together)
    float4 s[kMergeDepth];
    uint4 c[kMergeDepth];
    bool hit[kMergeDepth];
#pragma unroll
    for (int k = 0; k < kMergeDepth; ++k) {
      const int j = j0 + k < n ? j0 + k : n - 1;
      const Window w = widen(lwin[j]);
      const unsigned ux = (unsigned)(x - w.x0), uz = (unsigned)(zb - w.z0);
      hit[k] = j0 + k < n && ux < (unsigned)w.w && uz < (unsigned)w.h;
      const float* slab = slabs + (size_t)lpart[j] * a.slab_stride;
      s[k] = acc;
      c[k] = cnt;
      if (hit[k]) {
        s[k] = *reinterpret_cast<const float4*>(slab + (size_t)uz * w.w + ux);
        if (RED == kMean)
          c[k] = *reinterpret_cast<const uint4*>(slab + (size_t)w.w * w.h + (size_t)uz * w.w + ux);
      }
    }
#pragma unroll
    for (int k = 0; k < kMergeDepth; ++k) {
      if (!hit[k]) continue;
      if (RED == kMean) { cnt.x += c[k].x; cnt.y += c[k].y; cnt.z += c[k].z; cnt.w += c[k].w; }
      acc.x = combine<RED>(acc.x, s[k].x);
      acc.y = combine<RED>(acc.y, s[k].y);
      acc.z = combine<RED>(acc.z, s[k].z);
      acc.w = combine<RED>(acc.w, s[k].w);
    }
  }
  if (RED == kMean) mean_of(acc, cnt);
  const size_t cell = fo * (size_t)a.mh * a.mw + (size_t)zb * a.mw + x;
  *reinterpret_cast<float4*>(a.out + cell) = acc;
  const uint32_t mk = (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
                      ((uint32_t)mask_of(acc.z, a.fill) << 16) |
                      ((uint32_t)mask_of(acc.w, a.fill) << 24);
  *reinterpret_cast<uint32_t*>(a.mask + cell) = mk;
}

// Batch fuse (north_star "projected+fused"): fused[c] = max/min over the frames
// whose union window covers c of out[b][c]; frames that do not cover c hold the
// fill value there, so they cannot change the result and are never read.
struct FuseArgs {
  int B, dc, mh, mw;          // B frames in this launch, starting at frame b0
  int b0, accumulate;         // accumulate: fold into the current content of `fused`
  float fill;
  const Win16* unions;        // (B_total)   written by k_window_scatter
  const float* maps;          // (B_total, dc, mh, mw)
  float* fused;               // (dc, mh, mw)
  uint8_t* fused_mask;
};

// (16 groups x 16 lanes x 16 loads in flight: 8.0 us per launch at cfg4; 32 x 8 x 16: 8.9-9.2, 32 x 16: 8.2, 16 x 32: 9.2,
// 8 x 32: 9.8, 64 x 8: 11.3, 64 x 4 / 32 x 4 with 32 loads: 13.2 / 13.4 -- a tile's 100-200 candidates want many lanes)
constexpr int kFuseGroups = 16;     // k_fuse_windows: float4 groups of the fused map per block
constexpr int kFuseLanes = 16;       //   threads sharing one group (they split the candidate windows)

constexpr int kUnionGroups = 64;   // k_fuse_unions: float4 groups of the fused map per block,
constexpr int kUnionLanes = 4;     //   threads sharing one group (frames b = lane mod that many),
constexpr int kUnionDepth = 16;     //   map loads a thread keeps in flight
// Block = 64 groups x 4 frame lanes.  Each thread tests the unions of its frames
// (lane, lane + 4, ...) sixteen at a time, loads the covered maps (independent
// 16-byte loads), and the 4 partial results of a group are combined through LDS.
// (Round 4: 1 KB of a frame's map row per block instead of 512 B -- longer bursts per frame -- and
// all of a thread's 16 loads of a 64-frame batch in flight at once: 163 -> 136 us per cfg3 step
// (40 channels), 6.5 -> 5.7 us at cfg2; 32 x 8 x 8 before, 64 x 8, 128 x 2 / x 4, 64 x 2, 32 x 4
// measured worse.)
template <bool IS_MAX>
__global__ void __launch_bounds__(kUnionGroups * kUnionLanes)
k_fuse_unions(FuseArgs a) {
  __shared__ float4 part[kUnionLanes][kUnionGroups];
  __shared__ int4 lunion[kUnionGroups * kUnionLanes];      // union windows of up to 256 frames
  const int ch = blockIdx.y;
  const int g4 = a.mw >> 2;
  const int gi = threadIdx.x & (kUnionGroups - 1), lane = threadIdx.x / kUnionGroups;
  const int g = blockIdx.x * kUnionGroups + gi;
  const bool live = g < g4 * a.mh;
  const int z = live ? g / g4 : 0, x = live ? (g - z * g4) << 2 : 0;
  const size_t M = (size_t)a.mh * a.mw;
  const size_t cell = (size_t)z * a.mw + x;
  float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
  if (a.accumulate && lane == 0 && live)
    acc = *reinterpret_cast<const float4*>(a.fused + (size_t)ch * M + cell);
  constexpr int kStage = kUnionGroups * kUnionLanes;
  for (int c0 = 0; c0 < a.B; c0 += kStage) {
    // the union windows go through LDS: one load per thread instead of a dependent global
    // load in front of every map load
    if (c0 > 0) __syncthreads();
    if (c0 + (int)threadIdx.x < a.B) {
      const Window U = widen(a.unions[a.b0 + c0 + threadIdx.x]);
      lunion[threadIdx.x] = make_int4(U.x0, U.z0, U.w, U.h);
    }
    __syncthreads();
    const int n = a.B - c0 < kStage ? a.B - c0 : kStage;
    for (int b0 = lane; b0 < n; b0 += kUnionDepth * kUnionLanes) {
      float4 v[kUnionDepth];
#pragma unroll
      for (int k = 0; k < kUnionDepth; ++k) {
        const int bb = b0 + k * kUnionLanes;
        v[k] = acc;
        if (bb < n) {
          const int4 U = lunion[bb];
          if ((unsigned)(z - U.y) < (unsigned)U.w && (unsigned)(x - U.x) < (unsigned)U.z)
            v[k] = *reinterpret_cast<const float4*>(
                a.maps + ((size_t)(a.b0 + c0 + bb) * a.dc + ch) * M + cell);
        }
      }
#pragma unroll
      for (int k = 0; k < kUnionDepth; ++k) {
        acc.x = IS_MAX ? fmaxf(acc.x, v[k].x) : fminf(acc.x, v[k].x);
        acc.y = IS_MAX ? fmaxf(acc.y, v[k].y) : fminf(acc.y, v[k].y);
        acc.z = IS_MAX ? fmaxf(acc.z, v[k].z) : fminf(acc.z, v[k].z);
        acc.w = IS_MAX ? fmaxf(acc.w, v[k].w) : fminf(acc.w, v[k].w);
      }
    }
  }
  part[lane][gi] = acc;
  __syncthreads();
  if (lane == 0 && live) {
#pragma unroll
    for (int k = 1; k < kUnionLanes; ++k) {
      const float4 o = part[k][gi];
      acc.x = IS_MAX ? fmaxf(acc.x, o.x) : fminf(acc.x, o.x);
      acc.y = IS_MAX ? fmaxf(acc.y, o.y) : fminf(acc.y, o.y);
      acc.z = IS_MAX ? fmaxf(acc.z, o.z) : fminf(acc.z, o.z);
      acc.w = IS_MAX ? fmaxf(acc.w, o.w) : fminf(acc.w, o.w);
    }
    *reinterpret_cast<float4*>(a.fused + (size_t)ch * M + cell) = acc;
    const uint32_t mk = (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
                        ((uint32_t)mask_of(acc.z, a.fill) << 16) |
                        ((uint32_t)mask_of(acc.w, a.fill) << 24);
    *reinterpret_cast<uint32_t*>(a.fused_mask + (size_t)ch * M + cell) = mk;
  }
}

// Direct batch fuse from the slabs (dm_orth_project_fused_f32): fused[c] =
// max/min over every (frame, part) window covering c of its slab value -- the
// per-frame maps are never materialised.  Same block shape as k_fuse_unions; the
// kFuseLanes lanes of a group split the B * nparts windows.
struct FuseWinArgs {
  int nwin;                   // windows of this launch: (frames of the chunk) * nparts
  int b0;                     // first frame of the chunk
  int nparts, oc, ch0, oc_total, mh, mw;
  int slab_stride;
  int accumulate;
  int gx0, gz0, gx1, gz1;     // bounding box of every window of the call (cells, half open)
  float fill;
  const Win16* wins;          // (B_total, nparts)   written by k_window_scatter
  const float* slabs;         // ((b * oc + chl) * nparts + p) * slab_stride
  float* fused;               // (oc_total, mh, mw)
  uint8_t* fused_mask;
  const uint32_t* spans;      // (windows, span_rows) lo | hi << 16: the cells of each window row that its slab holds
  int span_rows;              //   (k_strip_fused flushes only those), or NULL: whole rows
};

constexpr int kFuseChunk = 1024;    // windows examined per candidate-list round
constexpr int kFuseDepth = 16;   // slab loads a lane keeps in flight

template <bool IS_MAX>
__global__ void __launch_bounds__(kFuseGroups * kFuseLanes)
k_fuse_windows(FuseWinArgs a) {
  __shared__ float4 part[kFuseLanes][kFuseGroups];
  __shared__ int4 cwin[kFuseChunk];     // candidate windows {x0, row of the window on the block's map row, w, span lo | hi << 16} ...
  __shared__ int cslab[kFuseChunk];     // ... and their slab index
  __shared__ int ncand;
  const int chl = blockIdx.y, ch = a.ch0 + chl;
  const int g4 = a.mw >> 2;
  const int total = g4 * a.mh;
  const size_t M = (size_t)a.mh * a.mw;
  // Most of a large global map is out of reach of the whole call.  The first `heavy`
  // blocks own the kFuseGroups-group tiles of the call's bounding box (one map row each); the
  // others stream the fill value (or, accumulating, only refresh the mask) over
  // everything outside it, 8 groups per thread: few, fat blocks instead of one tiny
  // block per 128 cells of a mostly empty map.
  const int bw4 = (a.gx1 - a.gx0) >> 2;                           // groups per bounding-box row
  const int tpr = (bw4 + kFuseGroups - 1) / kFuseGroups;          // tiles per bounding-box row
  const int heavy = a.gx1 > a.gx0 ? tpr * (a.gz1 - a.gz0) : 0;
  if ((int)blockIdx.x >= heavy) {                                 // block-uniform
    const int first = ((int)blockIdx.x - heavy) * (int)blockDim.x * 8 + (int)threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int g = first + k * (int)blockDim.x;
      if (g >= total) break;
      const int z = g / g4, x = (g - z * g4) << 2;
      if (z >= a.gz0 && z < a.gz1 && x >= a.gx0 && x < a.gx1) continue;     // a tile's cell
      const size_t cell = (size_t)ch * M + (size_t)z * a.mw + x;
      uint32_t mk = 0u;
      if (a.accumulate) {
        const float4 v = *reinterpret_cast<const float4*>(a.fused + cell);
        mk = (uint32_t)mask_of(v.x, a.fill) | ((uint32_t)mask_of(v.y, a.fill) << 8) |
             ((uint32_t)mask_of(v.z, a.fill) << 16) | ((uint32_t)mask_of(v.w, a.fill) << 24);
      } else {
        *reinterpret_cast<float4*>(a.fused + cell) = make_float4(a.fill, a.fill, a.fill, a.fill);
      }
      *reinterpret_cast<uint32_t*>(a.fused_mask + cell) = mk;
    }
    return;
  }
  const int gi = threadIdx.x & (kFuseGroups - 1), lane = threadIdx.x / kFuseGroups;
  const int trow = (int)blockIdx.x / tpr, tcol = (int)blockIdx.x - trow * tpr;
  const int z = a.gz0 + trow;
  const int gx = (a.gx0 >> 2) + tcol * kFuseGroups + gi;          // group index within the row
  const bool live = gx < (a.gx1 >> 2);
  const int x = (live ? gx : (a.gx0 >> 2)) << 2;
  const int z_lo = z, z_hi = z;
  const int x_lo = ((a.gx0 >> 2) + tcol * kFuseGroups) << 2;
  const int x_hi = x_lo + 4 * kFuseGroups < a.gx1 ? x_lo + 4 * kFuseGroups : a.gx1;
  const size_t cell = (size_t)ch * M + (size_t)z * a.mw + x;
  float4 acc = make_float4(a.fill, a.fill, a.fill, a.fill);
  if (a.accumulate && lane == 0 && live) acc = *reinterpret_cast<const float4*>(a.fused + cell);
  const int wlane = (int)threadIdx.x & 63;
  for (int c0 = 0; c0 < a.nwin; c0 += kFuseChunk) {
    // round 1: which windows of this chunk touch the block's cells at all?  One window per thread
    // and trip; a wave compacts its hits with a ballot and takes its share of the list with ONE
    // LDS atomic (one atomic per hit serialised the block on a single counter: on a trajectory a
    // tile has 100-200 candidates).
    if (threadIdx.x == 0) ncand = 0;
    __syncthreads();
    for (int r0 = c0; r0 < a.nwin && r0 < c0 + kFuseChunk; r0 += (int)blockDim.x) {     // (uniform trips)
      const int r = r0 + (int)threadIdx.x;
      bool hit = false;
      Window w = {0, 0, 0, 0};
      uint32_t span = 0u;
      if (r < a.nwin && r < c0 + kFuseChunk) {
        w = widen(a.wins[(size_t)a.b0 * a.nparts + r]);
        hit = w.w > 0 && w.z0 <= z_hi && w.z0 + w.h > z_lo && w.x0 < x_hi && w.x0 + w.w > x_lo;
        span = (uint32_t)w.x0 | ((uint32_t)(w.x0 + w.w) << 16);
        if (hit && a.spans) {      // (the block's cells all lie on map row z)
          span = a.spans[((size_t)a.b0 * a.nparts + r) * a.span_rows + (z - w.z0)];
          hit = (int)(span & 0xffffu) < x_hi && (int)(span >> 16) > x_lo;
        }
      }
      const unsigned long long hits = __builtin_amdgcn_ballot_w64(hit);
      if (hits != 0ull) {
        int base = 0;
        if (wlane == 0) base = atomicAdd(&ncand, __builtin_popcountll(hits));
        base = __builtin_amdgcn_readfirstlane(base);
        if (hit) {
          const int slot = base + __builtin_popcountll(hits & ((1ull << wlane) - 1ull));
          const int b = r / a.nparts, p = r - b * a.nparts;
          cwin[slot] = make_int4(w.x0, z - w.z0, w.w, (int)span);
          cslab[slot] = ((a.b0 + b) * a.oc + chl) * a.nparts + p;
        }
      }
    }
    __syncthreads();
    // round 2: the lanes of a group split the candidates, kFuseDepth slab loads in flight (16: a tile of a
    // trajectory has 100-200 candidates, 12-25 per lane -- with eight in flight that was up to four dependent
    // rounds of loads, 10.3 us per launch at cfg4; with sixteen 9.2)
    const int n = ncand;
    for (int i0 = lane; i0 < n; i0 += kFuseDepth * kFuseLanes) {
      float4 v[kFuseDepth];
#pragma unroll
      for (int k = 0; k < kFuseDepth; ++k) {
        const int i = i0 + k * kFuseLanes;
        v[k] = acc;
        if (i < n) {
          const int4 w = cwin[i];
          if (x >= (int)((unsigned)w.w & 0xffffu) && x < (int)((unsigned)w.w >> 16))
            v[k] = *reinterpret_cast<const float4*>(
                a.slabs + (size_t)cslab[i] * a.slab_stride + (size_t)w.y * w.z + (x - w.x));
        }
      }
#pragma unroll
      for (int k = 0; k < kFuseDepth; ++k) {
        acc.x = IS_MAX ? fmaxf(acc.x, v[k].x) : fminf(acc.x, v[k].x);
        acc.y = IS_MAX ? fmaxf(acc.y, v[k].y) : fminf(acc.y, v[k].y);
        acc.z = IS_MAX ? fmaxf(acc.z, v[k].z) : fminf(acc.z, v[k].z);
        acc.w = IS_MAX ? fmaxf(acc.w, v[k].w) : fminf(acc.w, v[k].w);
      }
    }
    __syncthreads();
  }
  part[lane][gi] = acc;
  __syncthreads();
  if (lane == 0 && live) {
#pragma unroll
    for (int k = 1; k < kFuseLanes; ++k) {
      const float4 o = part[k][gi];
      acc.x = IS_MAX ? fmaxf(acc.x, o.x) : fminf(acc.x, o.x);
      acc.y = IS_MAX ? fmaxf(acc.y, o.y) : fminf(acc.y, o.y);
      acc.z = IS_MAX ? fmaxf(acc.z, o.z) : fminf(acc.z, o.z);
      acc.w = IS_MAX ? fmaxf(acc.w, o.w) : fminf(acc.w, o.w);
    }
    *reinterpret_cast<float4*>(a.fused + cell) = acc;
    const uint32_t mk = (uint32_t)mask_of(acc.x, a.fill) | ((uint32_t)mask_of(acc.y, a.fill) << 8) |
                        ((uint32_t)mask_of(acc.z, a.fill) << 16) |
                        ((uint32_t)mask_of(acc.w, a.fill) << 24);
    *reinterpret_cast<uint32_t*>(a.fused_mask + cell) = mk;
  }
}

}  // namespace
}  // namespace dm
