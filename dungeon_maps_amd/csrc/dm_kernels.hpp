// Internal launcher declarations shared by the translation units of
// libdungeon_maps_amd.so.  The public surface is include/dungeon_maps_amd.h.
#pragma once
#include <hip/hip_runtime.h>

#include "dm_pixel.hpp"

namespace dm {

// dm_generic.hip -- fill / global-atomic scatter / finalize
size_t generic_workspace_bytes(const dm_params& p);
hipError_t run_generic(const dm_params& p, const dm_frame* frames_host, const float* depth,
                       const float* value, const uint8_t* valid, float* out,
                       uint8_t* mask, float* height, void* ws, hipStream_t s);
hipError_t run_generic_fused(const dm_params& p, const dm_frame* frames_host, const float* depth,
                             const float* value, const uint8_t* valid, float* out,
                             uint8_t* mask, int accumulate, void* ws, hipStream_t s);

hipError_t run_fuse_batch(const float* maps, int B, size_t n, float* out, bool is_max,
                          int accumulate, hipStream_t s);
hipError_t run_mask_from_map(const float* map, float fill, uint8_t* mask, size_t n,
                             hipStream_t s);

// dm_window.hip -- LDS-windowed scatter + slab merge (max/min; heights or value maps)
bool window_path_supported(const dm_params& p);
// Finite depth bounds for a call that lacks them (no pixel beyond them can land in the map);
// false: no such bounds.
bool bounded_depth_params(dm_params& p, const dm_frame* frames_host);
size_t window_workspace_bytes(const dm_params& p);
hipError_t run_window(const dm_params& p, const dm_frame* frames_host, const float* depth,
                      const float* value, const uint8_t* valid, float* out, uint8_t* mask,
                      float* height, float* fused, uint8_t* fused_mask, void* ws,
                      size_t ws_bytes, hipEvent_t before_projection, hipEvent_t after_projection,
                      hipStream_t s, const dm_frame* flow_frames_host = nullptr, float* flow_grid = nullptr,
                      bool* flow_done = nullptr);

hipError_t run_window_fused(const dm_params& p, const dm_frame* frames_host, const float* depth,
                            const float* value, const uint8_t* valid, float* out, uint8_t* mask,
                            int accumulate, void* ws, size_t ws_bytes, hipStream_t s);

// dm_strip.hip -- column strips with device-side geometry: owned cells straight to the map,
// shared cells through slabs (max / min).  hipErrorNotSupported: does not apply, nothing enqueued.
size_t strip_workspace_extra(const dm_params& p);
hipError_t run_strip(const dm_params& p, const dm_frame* frames_host, const float* depth,
                     const float* value, const uint8_t* valid, float* out, uint8_t* mask,
                     float* height, float* fused, uint8_t* fused_mask, void* ws, size_t ws_bytes,
                     int* status, hipEvent_t before_projection, hipEvent_t after_projection, hipStream_t s);

// dm_orth_project_fused_f32 on column strips (heights; groups of frames per workgroup)
hipError_t run_strip_fused(const dm_params& p, const dm_frame* frames_host, const float* depth,
                           const uint8_t* valid, float* out, uint8_t* mask, int accumulate, void* ws,
                           size_t ws_bytes, int* status, hipStream_t s);

size_t strip_prepared_bytes(const dm_params& p);
// hipErrorInvalidConfiguration: the plan differs from `must_match` (nothing enqueued)
hipError_t strip_prepare(const dm_params& p, const dm_frame* frames_host, void* prepared_dev,
                         size_t prepared_size, const dm_frames_plan* must_match, dm_frames_plan* plan_out,
                         hipStream_t s);
hipError_t run_strip_prepared(const dm_params& p, const dm_frames_plan& fp, const void* prepared_dev,
                              const float* depth, const float* value, const uint8_t* valid, float* out,
                              uint8_t* mask, float* height, float* fused, uint8_t* fused_mask, void* ws,
                              size_t ws_bytes, int* status, hipEvent_t before_projection,
                              hipEvent_t after_projection, hipStream_t s);

// dm_points.hip -- exact point-set primitives (affine, quantise, flat scatter)
hipError_t run_affine_points(const float* pts, const float* R, const float* t, int B,
                             size_t n_per_batch, int translate_first, float* out, hipStream_t s);
hipError_t run_map_quantize(const float* x, const float* z, const float* woff, const float* hoff,
                            int B, size_t n_per_batch, float res, int map_height, int flip,
                            long long* xb, long long* zb, hipStream_t s);
hipError_t run_camera_affine_grid(const dm_params& p, const dm_frame* frames_host,
                                  const float* depth, float* grid, void* ws, hipStream_t s);
hipError_t run_crop_interp(const float* src, const uint8_t* src_mask, const float* center, int B, int C, int h, int w,
                           int ch, int cw, float fill, int has_fill, int mode, float* dst, uint8_t* dst_mask,
                           hipStream_t s);
hipError_t run_crop_nearest(const float* src, const uint8_t* src_mask, const float* center, int B,
                            int C, int h, int w, int ch, int cw, float fill, int has_fill,
                            float* dst, uint8_t* dst_mask, hipStream_t s);
hipError_t run_fuse_bbox(const dm_fuse_src& s, int* stats, int init, hipStream_t st);
hipError_t run_fuse_bbox_multi(const dm_fuse_src* srcs, int n, int* stats, hipStream_t st);
hipError_t run_fuse_scatter_multi(const dm_fuse_src* srcs, int n, float woff, float hoff, int flip,
                                  int mh, int mw, int is_max, float* canvas, float* hcanvas,
                                  hipStream_t st);
hipError_t run_fuse_scatter(const dm_fuse_src& s, float woff, float hoff, int flip, int mh, int mw,
                            int is_max, float* canvas, float* hcanvas, hipStream_t st);
size_t scatter_workspace_bytes(size_t rows, size_t M, int has_fill, int reduction);
hipError_t run_scatter(const float* values, const long long* index, float* canvas, uint8_t* mask,
                       int R, int C, int Ci, size_t N, size_t M, float fill, int has_fill,
                       int reduction, void* ws, hipStream_t s);

}  // namespace dm
