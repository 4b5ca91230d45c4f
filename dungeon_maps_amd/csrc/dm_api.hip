// C ABI of libdungeon_maps_amd.so (include/dungeon_maps_amd.h): argument
// validation, path selection, error reporting.  No torch types, no allocation.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "dm_kernels.hpp"

namespace {

thread_local char g_err[512] = "";
thread_local int g_force_generic = 0;   // dm_debug_force_generic_path (tests)
thread_local hipEvent_t g_mid_event = nullptr;   // dm_debug_record_after_projection (bench)
thread_local hipEvent_t g_pre_event = nullptr;   // dm_debug_record_before_projection (bench)

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_params(const dm_params* p) {
  if (!p) return fail(DM_ERR_INVALID_ARGUMENT, "params is NULL");
  if (p->B < 0 || p->dc < 1 || p->vc < 0 || p->H < 1 || p->W < 1 || p->mh < 1 || p->mw < 1)
    return fail(DM_ERR_INVALID_ARGUMENT, "bad shape B=%d dc=%d vc=%d H=%d W=%d mh=%d mw=%d",
                p->B, p->dc, p->vc, p->H, p->W, p->mh, p->mw);
  if (p->vc && !(p->dc == 1 || p->dc == p->vc))
    return fail(DM_ERR_INVALID_ARGUMENT,
                "depth channels (%d) must be 1 or equal to value channels (%d)", p->dc, p->vc);
  if (!(p->valid_c == 0 || p->valid_c == 1 || p->valid_c == p->dc))
    return fail(DM_ERR_INVALID_ARGUMENT, "valid_c=%d must be 0, 1 or dc=%d", p->valid_c, p->dc);
  if (p->reduction < DM_REDUCE_MAX || p->reduction > DM_REDUCE_PROD)
    return fail(DM_ERR_INVALID_ARGUMENT, "unknown reduction %d", p->reduction);
  if ((int64_t)p->H * p->W >= (1ll << 31) || (int64_t)p->mh * p->mw >= (1ll << 31))
    return fail(DM_ERR_UNSUPPORTED, "image or map has more than 2^31 elements");
  if (p->B > 65535 || p->dc > 65535)
    return fail(DM_ERR_UNSUPPORTED, "B and dc are limited to 65535 per call");
  return DM_OK;
}

// Map widths that are not a multiple of 4 (odd ego-centric maps: 241 x 241 ...): the LDS-windowed
// path needs 16-byte rows.  Instead of the ~9x slower generic path, the call then projects into
// maps padded to the next multiple of 4 (workspace) and copies the real columns out: columns
// are independent, so the kept ones are identical.
inline bool padded_route(const dm_params& p) {
  if (p.mw % 4 == 0) return false;
  dm_params q = p;
  q.mw = (p.mw + 3) & ~3;
  return dm::window_path_supported(q);
}
inline size_t padded_bytes(const dm_params& p) {          // padded out + mask (+ height), 256-aligned
  const size_t mw4 = (size_t)((p.mw + 3) & ~3), oc = p.vc ? p.vc : p.dc;
  const size_t cells = (size_t)p.B * oc * p.mh * mw4, hcells = p.vc ? (size_t)p.B * p.dc * p.mh * mw4 : 0;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  return up(cells * 4) + up(cells) + up(hcells * 4);
}

__global__ void __launch_bounds__(256)
k_unpad(const float* __restrict__ src, const uint8_t* __restrict__ src_mask, float* __restrict__ dst,
        uint8_t* __restrict__ dst_mask, int mw, int mw4, size_t rows) {
  const size_t n = rows * (size_t)mw;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t r = i / mw, x = i - r * mw;
    dst[i] = src[r * mw4 + x];
    if (dst_mask) dst_mask[i] = src_mask[r * mw4 + x];
  }
}

}  // namespace

extern "C" {

int dm_version(void) { return DM_ABI_VERSION; }

// Builds that change results for timing's sake (the pixel loop without its arithmetic, the kernels without their
// fill duty or their LDS atomics: tools/experiments/measurement_switches_r4.patch) never ship: refused here.
#if defined(DM_X_NOMATH) || defined(DM_X_NOFILL2) || defined(DM_X_NOATOMIC) || defined(DM_X_COMBINE_NOP)
#error "a DM_X_NO* build computes wrong maps: apply it from tools/experiments, never to the product tree"
#endif
const char* dm_build_flags(void) {
  return ""
#ifdef DM_STAMPS
         "DM_STAMPS "
#endif
#ifdef DM_HOST_TIMING
         "DM_HOST_TIMING "
#endif
      ;
}

const char* dm_last_error(void) { return g_err; }

size_t dm_orth_project_workspace_bytes(const dm_params* p) {
  if (check_params(p) != DM_OK) return 0;
  size_t n = dm::generic_workspace_bytes(*p);
  if (dm::window_path_supported(*p)) {
    const size_t w = dm::window_workspace_bytes(*p) + dm::strip_workspace_extra(*p);
    if (w > n) n = w;
  } else if (padded_route(*p)) {
    dm_params q = *p;
    q.mw = (p->mw + 3) & ~3;
    const size_t w = padded_bytes(*p) + dm::window_workspace_bytes(q);
    if (w > n) n = w;
  }
  return n;
}

}  // extern "C"

// dm_orth_project_f32, and -- with `flow_frames` / `grid_dev` -- dm_orth_project_flow_f32: the same
// projection plus the ego-motion flow grid of the same depth maps, computed by the projection
// kernel itself where the window path's lean height kernel runs (one depth read for both), by the
// stand-alone kernel behind the projection otherwise.
static int orth_project_impl(const dm_params* p, const dm_frame* frames, const float* depth_dev,
                             const float* value_dev, const uint8_t* valid_dev, float* out_dev,
                             uint8_t* mask_dev, float* height_dev, float* fused_dev,
                             uint8_t* fused_mask_dev, void* workspace_dev, size_t workspace_bytes,
                             int32_t* status_dev, void* stream, const dm_frame* flow_frames,
                             float* grid_dev) {
  int rc = check_params(p);
  if (rc != DM_OK) return rc;
  if (p->B == 0) return DM_OK;
  if (!frames || !depth_dev || !out_dev || !mask_dev)
    return fail(DM_ERR_INVALID_ARGUMENT, "frames/depth/out/mask must not be NULL");
  if ((p->vc > 0) != (value_dev != nullptr))
    return fail(DM_ERR_INVALID_ARGUMENT, "value pointer and vc=%d disagree", p->vc);
  if ((p->valid_c > 0) != (valid_dev != nullptr))
    return fail(DM_ERR_INVALID_ARGUMENT, "valid pointer and valid_c=%d disagree", p->valid_c);
  if ((fused_dev != nullptr) != (fused_mask_dev != nullptr))
    return fail(DM_ERR_INVALID_ARGUMENT, "fused map and fused mask must be given together");
  if (fused_dev && p->reduction != DM_REDUCE_MAX && p->reduction != DM_REDUCE_MIN)
    return fail(DM_ERR_UNSUPPORTED, "the batch-fused map supports max/min only (got %d)",
                p->reduction);
  const size_t need = dm_orth_project_workspace_bytes(p);
  if (need > workspace_bytes || (need && !workspace_dev))
    return fail(DM_ERR_WORKSPACE_TOO_SMALL, "workspace %zu B < required %zu B", workspace_bytes,
                need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipErrorNotSupported;
  const hipEvent_t mid = g_mid_event, pre = g_pre_event;
  g_mid_event = nullptr;
  g_pre_event = nullptr;
  bool flowed = false;        // the projection kernel wrote the flow grid
  if (dm::window_path_supported(*p) && !g_force_generic) {
    // the reference's default is no depth truncation at all (maps.py:1267-1268): the fast paths
    // then run with bounds beyond which no ray can still be inside the map (same cells)
    dm_params q = *p;
    const dm_params& pp = dm::bounded_depth_params(q, frames) ? q : *p;
    e = dm::run_strip(pp, frames, depth_dev, value_dev, valid_dev, out_dev, mask_dev,
                      p->vc ? height_dev : nullptr, fused_dev, fused_mask_dev, workspace_dev,
                      workspace_bytes, status_dev, pre, mid, s);
    if (e == hipErrorNotSupported)
      e = dm::run_window(pp, frames, depth_dev, value_dev, valid_dev, out_dev, mask_dev,
                         p->vc ? height_dev : nullptr, fused_dev, fused_mask_dev, workspace_dev,
                         workspace_bytes, pre, mid, s, flow_frames, grid_dev, &flowed);
  } else if (padded_route(*p) && !g_force_generic &&
           reinterpret_cast<uintptr_t>(workspace_dev) % 256 == 0) {
    // project into padded maps at the head of the workspace, then copy the real columns out
    dm_params q = *p;
    q.mw = (p->mw + 3) & ~3;
    const size_t oc = p->vc ? p->vc : p->dc;
    const size_t cells = (size_t)p->B * oc * p->mh * q.mw;
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    unsigned char* base = static_cast<unsigned char*>(workspace_dev);
    float* out_pad = reinterpret_cast<float*>(base);
    uint8_t* mask_pad = base + up(cells * 4);
    float* height_pad = (p->vc && height_dev) ? reinterpret_cast<float*>(base + up(cells * 4) + up(cells))
                                              : nullptr;
    const size_t head = padded_bytes(*p);
    e = dm::run_window(q, frames, depth_dev, value_dev, valid_dev, out_pad, mask_pad, height_pad,
                       nullptr, nullptr, base + head, workspace_bytes - head, pre, nullptr, s);
    if (e == hipSuccess) {
      const size_t rows = (size_t)p->B * oc * p->mh;
      size_t blocks = (rows * p->mw + 1023) / 1024;
      if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL(k_unpad, dim3((unsigned)blocks), dim3(256), 0, s, out_pad, mask_pad, out_dev,
                         mask_dev, p->mw, q.mw, rows);
      if (height_pad) {
        const size_t hrows = (size_t)p->B * p->dc * p->mh;
        hipLaunchKernelGGL(k_unpad, dim3((unsigned)blocks), dim3(256), 0, s, height_pad,
                           (const uint8_t*)nullptr, height_dev, (uint8_t*)nullptr, p->mw, q.mw, hrows);
      }
      e = hipGetLastError();
      if (e == hipSuccess && mid) e = hipEventRecord(mid, s);
      if (e == hipSuccess && fused_dev) {
        const size_t n = oc * p->mh * p->mw;
        e = dm::run_fuse_batch(out_dev, p->B, n, fused_dev, p->reduction == DM_REDUCE_MAX, 0, s);
        if (e == hipSuccess) e = dm::run_mask_from_map(fused_dev, p->fill, fused_mask_dev, n, s);
      }
    }
  }
  if (e == hipErrorNotSupported) { // nothing enqueued: a window exceeds LDS, odd alignment, ...
    if (pre) (void)hipEventRecord(pre, s);
    e = dm::run_generic(*p, frames, depth_dev, value_dev, valid_dev, out_dev, mask_dev,
                        p->vc ? height_dev : nullptr, workspace_dev, s);
    if (e == hipSuccess && mid) e = hipEventRecord(mid, s);
    if (e == hipSuccess && fused_dev) {
      const size_t n = (size_t)(p->vc ? p->vc : p->dc) * p->mh * p->mw;
      e = dm::run_fuse_batch(out_dev, p->B, n, fused_dev, p->reduction == DM_REDUCE_MAX, 0, s);
      if (e == hipSuccess) e = dm::run_mask_from_map(fused_dev, p->fill, fused_mask_dev, n, s);
    }
  }
  if (e == hipSuccess && grid_dev && !flowed)       // (no fused kernel for this call: the stand-alone one)
    e = dm::run_camera_affine_grid(*p, flow_frames, depth_dev, grid_dev, workspace_dev, s);
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

extern "C" {

int dm_orth_project_f32(const dm_params* p, const dm_frame* frames, const float* depth_dev,
                        const float* value_dev, const uint8_t* valid_dev, float* out_dev,
                        uint8_t* mask_dev, float* height_dev, float* fused_dev,
                        uint8_t* fused_mask_dev, void* workspace_dev, size_t workspace_bytes,
                        int32_t* status_dev, void* stream) {
  return orth_project_impl(p, frames, depth_dev, value_dev, valid_dev, out_dev, mask_dev, height_dev, fused_dev,
                           fused_mask_dev, workspace_dev, workspace_bytes, status_dev, stream, nullptr, nullptr);
}

int dm_orth_project_flow_f32(const dm_params* p, const dm_frame* frames, const dm_frame* flow_frames,
                             const float* depth_dev, const uint8_t* valid_dev, float* out_dev,
                             uint8_t* mask_dev, float* fused_dev, uint8_t* fused_mask_dev, float* grid_dev,
                             void* workspace_dev, size_t workspace_bytes, int32_t* status_dev, void* stream) {
  if (!p || p->vc != 0)
    return fail(DM_ERR_INVALID_ARGUMENT, "dm_orth_project_flow_f32 projects heights (vc = 0)");
  if (!flow_frames || !grid_dev)
    return fail(DM_ERR_INVALID_ARGUMENT, "flow_frames / grid must not be NULL");
  if (p->B == 0) return DM_OK;
  return orth_project_impl(p, frames, depth_dev, nullptr, valid_dev, out_dev, mask_dev, nullptr, fused_dev,
                           fused_mask_dev, workspace_dev, workspace_bytes, status_dev, stream, flow_frames, grid_dev);
}

int dm_frames_fill_f32(const float* static_table, int32_t B, const float* pose, const float* sin_yaw,
                       const float* cos_yaw, float* out_table) {
  if (!static_table || !pose || !sin_yaw || !cos_yaw || !out_table || B < 0)
    return fail(DM_ERR_INVALID_ARGUMENT, "dm_frames_fill_f32: NULL argument or negative B");
  static_assert(sizeof(dm_frame) == 32 * sizeof(float), "dm_frame is 32 floats");
  memcpy(out_table, static_table, (size_t)B * sizeof(dm_frame));
  dm_frame* f = reinterpret_cast<dm_frame*>(out_table);
  for (int32_t b = 0; b < B; ++b) {
    const float yaw = pose[3 * b + 2];
    // utils.py:323-324: |angle| <= 1e-3 -> 0, i.e. sin = 0 and cos = 1 exactly
    const bool small = (yaw < 0.0f ? -yaw : yaw) <= 0.001f;
    const float s = small ? 0.0f : sin_yaw[b];
    const float c = small ? 1.0f : cos_yaw[b];
    const float d = 1.0f - (1.0f - c);          // (1 + sin * 0) + (1 - cos) * (-1), float32 (utils.py:326)
    f[b].Ry[0] = d; f[b].Ry[2] = s; f[b].Ry[6] = -s; f[b].Ry[8] = d;
    f[b].tx = pose[3 * b]; f[b].tz = pose[3 * b + 1];
  }
  return DM_OK;
}

size_t dm_frames_prepared_bytes(const dm_params* p) {
  if (check_params(p) != DM_OK || p->B < 1) return 0;
  return dm::strip_prepared_bytes(*p);
}

int dm_frames_prepare_f32(const dm_params* p, const dm_frame* frames, void* prepared_dev,
                          size_t prepared_bytes, const dm_frames_plan* must_match,
                          dm_frames_plan* plan_out, void* stream) {
  int rc = check_params(p);
  if (rc != DM_OK) return rc;
  if (!frames || !prepared_dev || !plan_out || p->B < 1)
    return fail(DM_ERR_INVALID_ARGUMENT, "frames/prepared/plan must not be NULL, B >= 1");
  const hipError_t e = dm::strip_prepare(*p, frames, prepared_dev, prepared_bytes, must_match, plan_out,
                                         static_cast<hipStream_t>(stream));
  if (e == hipErrorNotSupported)
    return fail(DM_ERR_UNSUPPORTED, "these parameters / frames cannot be prepared (the strip path does "
                                    "not apply): use dm_orth_project_f32");
  if (e == hipErrorInvalidConfiguration)
    return fail(DM_ERR_PLAN_MISMATCH, "the frames need a different launch plan than `must_match`; "
                                      "nothing was uploaded");
  if (e == hipErrorInvalidValue)
    return fail(DM_ERR_WORKSPACE_TOO_SMALL, "prepared buffer: %zu B given, %zu B (256-byte aligned) needed",
                prepared_bytes, dm::strip_prepared_bytes(*p));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP copy failed: %s", hipGetErrorString(e));
  return DM_OK;
}

int dm_orth_project_prepared_f32(const dm_params* p, const dm_frames_plan* plan, const void* prepared_dev,
                                 const float* depth_dev, const float* value_dev,
                                 const uint8_t* valid_dev, float* out_dev, uint8_t* mask_dev,
                                 float* height_dev, float* fused_dev, uint8_t* fused_mask_dev,
                                 void* workspace_dev, size_t workspace_bytes, int32_t* status_dev,
                                 void* stream) {
  int rc = check_params(p);
  if (rc != DM_OK) return rc;
  if (!plan || !prepared_dev || !depth_dev || !out_dev || !mask_dev || p->B < 1)
    return fail(DM_ERR_INVALID_ARGUMENT, "plan/prepared/depth/out/mask must not be NULL, B >= 1");
  if ((p->vc > 0) != (value_dev != nullptr))
    return fail(DM_ERR_INVALID_ARGUMENT, "value pointer and vc=%d disagree", p->vc);
  if ((p->valid_c > 0) != (valid_dev != nullptr))
    return fail(DM_ERR_INVALID_ARGUMENT, "valid pointer and valid_c=%d disagree", p->valid_c);
  if ((fused_dev != nullptr) != (fused_mask_dev != nullptr))
    return fail(DM_ERR_INVALID_ARGUMENT, "fused map and fused mask must be given together");
  const size_t need = dm_orth_project_workspace_bytes(p);
  if (need > workspace_bytes || !workspace_dev)
    return fail(DM_ERR_WORKSPACE_TOO_SMALL, "workspace %zu B < required %zu B", workspace_bytes, need);
  const hipEvent_t mid = g_mid_event, pre = g_pre_event;
  g_mid_event = nullptr;
  g_pre_event = nullptr;
  const hipError_t e = dm::run_strip_prepared(*p, *plan, prepared_dev, depth_dev, value_dev, valid_dev,
                                              out_dev, mask_dev, p->vc ? height_dev : nullptr, fused_dev,
                                              fused_mask_dev, workspace_dev, workspace_bytes, status_dev,
                                              pre, mid, static_cast<hipStream_t>(stream));
  if (e == hipErrorNotSupported)
    return fail(DM_ERR_UNSUPPORTED, "plan does not match the parameters, or misaligned pointers");
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

int dm_orth_project_fused_f32(const dm_params* p, const dm_frame* frames,
                              const float* depth_dev, const float* value_dev,
                              const uint8_t* valid_dev, float* out_dev, uint8_t* mask_dev,
                              int accumulate, void* workspace_dev, size_t workspace_bytes,
                              int32_t* status_dev, void* stream) {
  int rc = check_params(p);
  if (rc != DM_OK) return rc;
  if (p->reduction != DM_REDUCE_MAX && p->reduction != DM_REDUCE_MIN)
    return fail(DM_ERR_UNSUPPORTED, "fused projection supports max/min only (got %d)",
                p->reduction);
  if (!out_dev || !mask_dev || (p->B > 0 && (!frames || !depth_dev)))
    return fail(DM_ERR_INVALID_ARGUMENT, "frames/depth/out/mask must not be NULL");
  const size_t need = dm_orth_project_workspace_bytes(p);
  if (need > workspace_bytes || !workspace_dev)
    return fail(DM_ERR_WORKSPACE_TOO_SMALL, "workspace %zu B < required %zu B", workspace_bytes,
                need);
  if ((p->vc > 0) != (value_dev != nullptr) && p->B > 0)
    return fail(DM_ERR_INVALID_ARGUMENT, "value pointer and vc=%d disagree", p->vc);
  if ((p->valid_c > 0) != (valid_dev != nullptr) && p->B > 0)
    return fail(DM_ERR_INVALID_ARGUMENT, "valid pointer and valid_c=%d disagree", p->valid_c);
  if (p->B == 0 && !accumulate)
    return fail(DM_ERR_INVALID_ARGUMENT, "cannot fuse an empty batch without accumulate");
  hipError_t e = hipErrorNotSupported;
  if (p->B > 0 && dm::window_path_supported(*p) && !g_force_generic)
    e = dm::run_strip_fused(*p, frames, depth_dev, valid_dev, out_dev, mask_dev, accumulate, workspace_dev,
                            workspace_bytes, status_dev, static_cast<hipStream_t>(stream));
  if (e == hipErrorNotSupported && p->B > 0 && dm::window_path_supported(*p) && !g_force_generic)
    e = dm::run_window_fused(*p, frames, depth_dev, value_dev, valid_dev, out_dev, mask_dev,
                             accumulate, workspace_dev, workspace_bytes,
                             static_cast<hipStream_t>(stream));
  if (e == hipErrorNotSupported)
    e = dm::run_generic_fused(*p, frames, depth_dev, value_dev, valid_dev, out_dev, mask_dev,
                              accumulate, workspace_dev, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

int dm_camera_affine_grid_f32(const dm_params* p, const dm_frame* frames, const float* depth_dev,
                              float* grid_dev, void* workspace_dev, size_t workspace_bytes,
                              void* stream) {
  if (!p) return fail(DM_ERR_INVALID_ARGUMENT, "params is NULL");
  if (p->B < 0 || p->dc < 1 || p->H < 1 || p->W < 1 || p->B > 65535 || p->dc > 65535 ||
      (int64_t)p->H * p->W >= (1ll << 31))
    return fail(DM_ERR_INVALID_ARGUMENT, "bad shape B=%d dc=%d H=%d W=%d", p->B, p->dc, p->H, p->W);
  if (p->B == 0) return DM_OK;
  if (!frames || !depth_dev || !grid_dev)
    return fail(DM_ERR_INVALID_ARGUMENT, "frames/depth/grid must not be NULL");
  if (!workspace_dev || workspace_bytes < (size_t)p->B * sizeof(dm_frame))
    return fail(DM_ERR_WORKSPACE_TOO_SMALL, "workspace %zu B < required %zu B", workspace_bytes,
                (size_t)p->B * sizeof(dm_frame));
  hipError_t e = dm::run_camera_affine_grid(*p, frames, depth_dev, grid_dev, workspace_dev,
                                            static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

static int check_fuse_src(const dm_fuse_src* s) {
  if (!s) return fail(DM_ERR_INVALID_ARGUMENT, "src is NULL");
  if (s->b < 0 || s->b > DM_FUSE_MAX_BATCH || s->c < 0 || s->c > 65535 || s->h < 1 || s->w < 1 ||
      (int64_t)s->h * s->w >= (1ll << 31) || (s->hc != 1 && s->hc != s->c) ||
      (s->mc != 1 && s->mc != s->c))
    return fail(DM_ERR_INVALID_ARGUMENT, "bad source map b=%d c=%d hc=%d mc=%d %dx%d", s->b, s->c,
                s->hc, s->mc, s->h, s->w);
  if (s->b && s->c && (!s->height_dev || !s->mask_dev))
    return fail(DM_ERR_INVALID_ARGUMENT, "height/mask must not be NULL");
  return DM_OK;
}

int dm_fuse_bbox_f32(const dm_fuse_src* src, int32_t* stats_dev, int init, void* stream) {
  const int rc = check_fuse_src(src);
  if (rc != DM_OK) return rc;
  if (!stats_dev) return fail(DM_ERR_INVALID_ARGUMENT, "stats is NULL");
  if (src->b == 0 || src->c == 0)
    return fail(DM_ERR_INVALID_ARGUMENT, "empty source map");
  const dm_fuse_src& s = *src;
  hipError_t e = hipSuccess;
  e = dm::run_fuse_bbox(s, stats_dev, init, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

int dm_fuse_scatter_f32(const dm_fuse_src* src, float width_offset, float height_offset, int flip_h,
                        int64_t map_height, int64_t map_width, int reduction, float* canvas_dev,
                        float* height_canvas_dev, void* stream) {
  const int rc = check_fuse_src(src);
  if (rc != DM_OK) return rc;
  if (reduction != DM_REDUCE_MAX && reduction != DM_REDUCE_MIN)
    return fail(DM_ERR_UNSUPPORTED, "fused map scatter supports max/min only (got %d)", reduction);
  if (map_height < 1 || map_width < 1 || map_height * map_width >= (1ll << 31))
    return fail(DM_ERR_INVALID_ARGUMENT, "bad target size %lldx%lld", (long long)map_height,
                (long long)map_width);
  if (src->b == 0 || src->c == 0) return DM_OK;
  if (!canvas_dev) return fail(DM_ERR_INVALID_ARGUMENT, "canvas is NULL");
  hipError_t e = dm::run_fuse_scatter(*src, width_offset, height_offset, flip_h, (int)map_height,
                                      (int)map_width, reduction == DM_REDUCE_MAX, canvas_dev,
                                      height_canvas_dev, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

static int check_fuse_sources(const dm_fuse_src* srcs, int32_t n) {
  if (!srcs || n < 1 || n > DM_FUSE_MAX_SOURCES)
    return fail(DM_ERR_INVALID_ARGUMENT, "1 .. %d source maps per call (got %d)", DM_FUSE_MAX_SOURCES, (int)n);
  for (int i = 0; i < n; ++i) {
    const int rc = check_fuse_src(srcs + i);
    if (rc != DM_OK) return rc;
    if (srcs[i].b != srcs[0].b || srcs[i].c != srcs[0].c)
      return fail(DM_ERR_INVALID_ARGUMENT, "source maps of one call share b and c (%d: b=%d c=%d against b=%d c=%d)",
                  i, srcs[i].b, srcs[i].c, srcs[0].b, srcs[0].c);
  }
  if (srcs[0].b == 0 || srcs[0].c == 0) return fail(DM_ERR_INVALID_ARGUMENT, "empty source maps");
  if ((int64_t)srcs[0].b * n > 65535) return fail(DM_ERR_INVALID_ARGUMENT, "too many batch rows");
  return DM_OK;
}

int dm_fuse_bbox_multi_f32(const dm_fuse_src* srcs, int32_t n, int32_t* stats_dev, void* stream) {
  const int rc = check_fuse_sources(srcs, n);
  if (rc != DM_OK) return rc;
  if (!stats_dev) return fail(DM_ERR_INVALID_ARGUMENT, "stats is NULL");
  const hipError_t e = dm::run_fuse_bbox_multi(srcs, n, stats_dev, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

int dm_fuse_bbox_read_i32(const int32_t* stats_dev, int32_t* stats_host, void* stream) {
  if (!stats_dev || !stats_host) return fail(DM_ERR_INVALID_ARGUMENT, "stats must not be NULL");
  // one pinned staging block per host thread and GPU (a copy into pageable memory goes through the runtime's own
  // staging and costs about twice the time)
  static thread_local int32_t* pinned[8] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= 8) return fail(DM_ERR_LAUNCH, "hipGetDevice failed");
  if (!pinned[dev]) {
    e = hipHostMalloc(reinterpret_cast<void**>(&pinned[dev]), 64, hipHostMallocDefault);
    if (e != hipSuccess) { pinned[dev] = nullptr; return fail(DM_ERR_LAUNCH, "hipHostMalloc failed: %s", hipGetErrorString(e)); }
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  e = hipMemcpyAsync(pinned[dev], stats_dev, 5 * sizeof(int32_t), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "reading the bounding box failed: %s", hipGetErrorString(e));
  memcpy(stats_host, pinned[dev], 5 * sizeof(int32_t));
  return DM_OK;
}

int dm_fuse_scatter_multi_f32(const dm_fuse_src* srcs, int32_t n, float width_offset, float height_offset,
                              int flip_h, int64_t map_height, int64_t map_width, int reduction,
                              float* canvas_dev, float* height_canvas_dev, void* stream) {
  const int rc = check_fuse_sources(srcs, n);
  if (rc != DM_OK) return rc;
  if (reduction != DM_REDUCE_MAX && reduction != DM_REDUCE_MIN)
    return fail(DM_ERR_UNSUPPORTED, "fused map scatter supports max/min only (got %d)", reduction);
  if (map_height < 1 || map_width < 1 || map_height * map_width >= (1ll << 31))
    return fail(DM_ERR_INVALID_ARGUMENT, "bad target size %lldx%lld", (long long)map_height,
                (long long)map_width);
  if (!canvas_dev) return fail(DM_ERR_INVALID_ARGUMENT, "canvas is NULL");
  const hipError_t e = dm::run_fuse_scatter_multi(srcs, n, width_offset, height_offset, flip_h, (int)map_height,
                                                  (int)map_width, reduction == DM_REDUCE_MAX, canvas_dev,
                                                  height_canvas_dev, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

int dm_crop_nearest_f32(const float* image_dev, const uint8_t* mask_dev, const float* center_dev,
                        int64_t B, int64_t C, int64_t h, int64_t w, int64_t crop_h, int64_t crop_w,
                        float fill, int has_fill, float* out_dev, uint8_t* out_mask_dev,
                        void* stream) {
  if (B < 0 || C < 0 || B > 65535 || C > 65535 || h < 1 || w < 1 || crop_h < 0 || crop_w < 0 ||
      h > (1 << 23) - 4 || w > (1 << 23) - 4 || crop_h * crop_w >= (1ll << 31) || h * w >= (1ll << 31))
    return fail(DM_ERR_INVALID_ARGUMENT, "bad shape B=%lld C=%lld %lldx%lld -> %lldx%lld",
                (long long)B, (long long)C, (long long)h, (long long)w, (long long)crop_h,
                (long long)crop_w);
  if (B == 0 || C == 0 || crop_h == 0 || crop_w == 0) return DM_OK;
  if (!image_dev || !center_dev || !out_dev || (mask_dev != nullptr) != (out_mask_dev != nullptr))
    return fail(DM_ERR_INVALID_ARGUMENT, "image/center/out must not be NULL; mask and out_mask go together");
  hipError_t e = dm::run_crop_nearest(image_dev, mask_dev, center_dev, (int)B, (int)C, (int)h, (int)w,
                                      (int)crop_h, (int)crop_w, fill, has_fill, out_dev,
                                      out_mask_dev, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

int dm_crop_sample_f32(const float* image_dev, const uint8_t* mask_dev, const float* center_dev,
                       int64_t B, int64_t C, int64_t h, int64_t w, int64_t crop_h, int64_t crop_w,
                       float fill, int has_fill, int mode, float* out_dev, uint8_t* out_mask_dev,
                       void* stream) {
  if (mode == DM_SAMPLE_NEAREST)
    return dm_crop_nearest_f32(image_dev, mask_dev, center_dev, B, C, h, w, crop_h, crop_w, fill, has_fill, out_dev,
                               out_mask_dev, stream);
  if (mode != DM_SAMPLE_BILINEAR && mode != DM_SAMPLE_BICUBIC)
    return fail(DM_ERR_INVALID_ARGUMENT, "sampling mode %d", mode);
  if (B < 0 || C < 0 || B > 65535 || C > 65535 || h < 1 || w < 1 || crop_h < 0 || crop_w < 0 ||
      h > (1 << 23) - 4 || w > (1 << 23) - 4 || crop_h * crop_w >= (1ll << 31) || h * w >= (1ll << 31))
    return fail(DM_ERR_INVALID_ARGUMENT, "bad shape B=%lld C=%lld %lldx%lld -> %lldx%lld",
                (long long)B, (long long)C, (long long)h, (long long)w, (long long)crop_h,
                (long long)crop_w);
  if (B == 0 || C == 0 || crop_h == 0 || crop_w == 0) return DM_OK;
  if (!image_dev || !center_dev || !out_dev || (mask_dev != nullptr) != (out_mask_dev != nullptr))
    return fail(DM_ERR_INVALID_ARGUMENT, "image/center/out must not be NULL; mask and out_mask go together");
  hipError_t e = dm::run_crop_interp(image_dev, mask_dev, center_dev, (int)B, (int)C, (int)h, (int)w, (int)crop_h,
                                     (int)crop_w, fill, has_fill, mode, out_dev, out_mask_dev,
                                     static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

int dm_affine_points_f32(const float* pts_dev, const float* R_dev, const float* t_dev, int64_t B,
                         size_t n, int translate_first, float* out_dev, void* stream) {
  if (B < 0 || B > 65535) return fail(DM_ERR_INVALID_ARGUMENT, "bad batch %lld", (long long)B);
  if (B == 0 || n == 0) return DM_OK;
  if (!pts_dev || !R_dev || !t_dev || !out_dev)
    return fail(DM_ERR_INVALID_ARGUMENT, "pts/R/t/out must not be NULL");
  hipError_t e = dm::run_affine_points(pts_dev, R_dev, t_dev, (int)B, n, translate_first, out_dev,
                                       static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

int dm_map_quantize_f32(const float* x_dev, const float* z_dev, const float* woff_dev,
                        const float* hoff_dev, int64_t B, size_t n, float res, int32_t map_height,
                        int32_t flip_h, int64_t* xb_dev, int64_t* zb_dev, void* stream) {
  if (B < 0 || B > 65535) return fail(DM_ERR_INVALID_ARGUMENT, "bad batch %lld", (long long)B);
  if (B == 0 || n == 0) return DM_OK;
  if (!x_dev || !z_dev || !woff_dev || !hoff_dev || !xb_dev || !zb_dev)
    return fail(DM_ERR_INVALID_ARGUMENT, "x/z/offsets/outputs must not be NULL");
  hipError_t e = dm::run_map_quantize(x_dev, z_dev, woff_dev, hoff_dev, (int)B, n, res, map_height,
                                      flip_h != 0, reinterpret_cast<long long*>(xb_dev),
                                      reinterpret_cast<long long*>(zb_dev),
                                      static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

size_t dm_scatter_workspace_bytes(int64_t R, int32_t C, size_t M, int32_t has_fill,
                                  int32_t reduction) {
  if (R <= 0 || C <= 0) return 0;
  return dm::scatter_workspace_bytes((size_t)R * C, M, has_fill, reduction);
}

int dm_scatter_f32(const float* values_dev, const int64_t* index_dev, float* canvas_dev,
                   uint8_t* mask_dev, int64_t R, int32_t C, int32_t Ci, size_t N, size_t M,
                   float fill, int32_t has_fill, int32_t reduction, void* workspace_dev,
                   size_t workspace_bytes, void* stream) {
  if (R < 0 || C < 1 || !(Ci == 1 || Ci == C))
    return fail(DM_ERR_INVALID_ARGUMENT, "bad shape R=%lld C=%d Ci=%d", (long long)R, C, Ci);
  if (reduction < DM_REDUCE_MAX || reduction > DM_REDUCE_PROD)
    return fail(DM_ERR_INVALID_ARGUMENT, "unknown reduction %d", reduction);
  if ((int64_t)R * C > 65535) return fail(DM_ERR_UNSUPPORTED, "R*C is limited to 65535 rows");
  if (R == 0 || M == 0) return DM_OK;
  if (!canvas_dev || !mask_dev || (N > 0 && (!values_dev || !index_dev)))
    return fail(DM_ERR_INVALID_ARGUMENT, "values/index/canvas/mask must not be NULL");
  const size_t need = dm_scatter_workspace_bytes(R, C, M, has_fill, reduction);
  if (need > workspace_bytes || (need && !workspace_dev))
    return fail(DM_ERR_WORKSPACE_TOO_SMALL, "workspace %zu B < required %zu B", workspace_bytes,
                need);
  hipError_t e = dm::run_scatter(values_dev, reinterpret_cast<const long long*>(index_dev),
                                 canvas_dev, mask_dev, (int)R, C, Ci, N, M, fill, has_fill != 0,
                                 reduction, workspace_dev, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

void dm_debug_record_before_projection(void* event) {
  g_pre_event = static_cast<hipEvent_t>(event);
}

void dm_debug_record_after_projection(void* event) {
  g_mid_event = static_cast<hipEvent_t>(event);
}

int dm_debug_force_generic_path(int on) {
  const int old = g_force_generic;
  g_force_generic = on != 0;
  return old;
}

int dm_fuse_batch_f32(const float* maps_dev, int64_t B, size_t n, float* out_dev, int reduction,
                      int accumulate, void* stream) {
  if (reduction != DM_REDUCE_MAX && reduction != DM_REDUCE_MIN)
    return fail(DM_ERR_UNSUPPORTED, "fuse supports max/min only (got %d)", reduction);
  if (B < 0 || B > 0x7fffffff) return fail(DM_ERR_INVALID_ARGUMENT, "bad batch %lld", (long long)B);
  if (n == 0 || (B == 0 && accumulate)) return DM_OK;
  if (B == 0) return fail(DM_ERR_INVALID_ARGUMENT, "cannot fuse an empty batch without accumulate");
  if (!maps_dev || !out_dev) return fail(DM_ERR_INVALID_ARGUMENT, "maps/out must not be NULL");
  hipError_t e = dm::run_fuse_batch(maps_dev, (int)B, n, out_dev, reduction == DM_REDUCE_MAX,
                                    accumulate, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

int dm_mask_from_map_f32(const float* map_dev, float fill, uint8_t* mask_dev, size_t n,
                         void* stream) {
  if (n == 0) return DM_OK;
  if (!map_dev || !mask_dev) return fail(DM_ERR_INVALID_ARGUMENT, "map/mask must not be NULL");
  hipError_t e = dm::run_mask_from_map(map_dev, fill, mask_dev, n,
                                       static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(DM_ERR_LAUNCH, "HIP launch failed: %s", hipGetErrorString(e));
  return DM_OK;
}

}  // extern "C"
