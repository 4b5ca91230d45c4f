/*
 * dungeon_maps_amd_debug.h -- test and measurement hooks of libdungeon_maps_amd.so.
 *
 * NOT part of the drop-in boundary (include/dungeon_maps_amd.h): nothing a caller of the projector needs.
 * The GPU tests, bench.py and tools/ use these to force a path / a split, to read back which path a call
 * took, to evaluate the host-side geometry without a GPU, and to place measurement events.  Every switch is
 * per calling thread and changes WHICH exact path computes a result, never the result.
 */
#ifndef DUNGEON_MAPS_AMD_DEBUG_H
#define DUNGEON_MAPS_AMD_DEBUG_H

#include "dungeon_maps_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Test hook: non-zero forces dm_orth_project_f32 onto the generic
 * (global-atomic) path on the calling thread; returns the previous setting.
 * The LDS-windowed fast path is checked against it on the device.
 */
int dm_debug_force_generic_path(int on);

/*
 * Test hook: how the LDS-windowed path split the frames of the calling thread's most recent
 * dm_orth_project_f32 / dm_orth_project_fused_f32 call: out4 = {image part columns, image
 * part rows, depth bands, times the frames went through in halves since the last query};
 * parts 0 x 0 x 0 when the windows did not fit and the call took the generic path.
 */
void dm_debug_last_split(int32_t* out4);

/*
 * Test hook, host only (no GPU needed): the split of the image dm_orth_project_f32 derives for
 * `p` with at least min_parts image parts and pd depth bands (1..8), and the map window of
 * every (frame, part) for the given poses -- the bound the LDS-windowed path relies on: a
 * pixel of a part can only land inside the part's window.
 *   out_parts[5]  = {strips pc, row blocks pr, depth bands pd, strip width wp, block height hp}
 *   out_windows   (B, pc*pr*pd, 4) int32 {x0, z0, w, h}, part index (band*pr + row)*pc + strip;
 *                 NULL: only the split.  window_capacity = windows the buffer holds.
 * Returns the number of parts per frame, negative on bad arguments.
 */
int dm_debug_windows(const dm_params* p, const dm_frame* frames, int min_parts, int pd,
                     int32_t* out_parts, int32_t* out_windows, size_t window_capacity);

/*
 * Test hook: non-zero makes the calling thread's projections take a split with depth bands
 * whenever one fits in LDS, also where the cost model would choose the generic path (small
 * images); returns the previous setting.  Lets the parity tests cover bands on small shapes.
 */
int dm_debug_force_bands(int on);

/*
 * Test hooks of the strip path (dm_strip.hip: column strips whose map windows are derived on
 * the device; cells only one strip can reach go straight to the map).
 *   dm_debug_last_path           which path the calling thread's last projection took:
 *                                0 generic (global atomics), 1 LDS windows with host geometry,
 *                                2 strip path.
 *   dm_debug_force_legacy_window non-zero keeps the calling thread's projections off the strip
 *                                path (they take path 1 or 0); returns the previous setting.
 *   dm_debug_force_strips        1..8: the calling thread's projections are cut into this many
 *                                column strips whatever the cost model says (0 = back to the
 *                                model); returns the previous setting.  Lets the tests run the
 *                                strip path on small images.
 *   dm_debug_fill_split          where the strip path's projections of the calling thread store the
 *                                fill value of the map rows outside a frame's union window: -1 (the
 *                                default) under the pixel loop with the rest of the fill duty; 0..8
 *                                (measurement settings) out of the loop -- that many of every eight
 *                                such rows of a wave in the scatter kernel's head, the others in the
 *                                combine kernel.  Returns the previous setting.  Same results.
 *   dm_debug_force_nt_fill       which cache policy the strip path's fill stores of the calling
 *                                thread's projections take: 1 non-temporal, 0 the default policy,
 *                                -1 the library's own rule (non-temporal where the call's batch
 *                                fuse follows or the maps exceed what the Infinity Cache keeps);
 *                                returns the previous setting.  Same results; lets a measurement
 *                                time a plain call on the kernel variant a fused call launches.
 *   dm_debug_strip_value_list    0: value maps of three channels or more recompute every pixel's cell
 *                                per channel on the strip path of the calling thread, instead of
 *                                taking it from the list the index pass leaves (non-zero, the
 *                                default); returns the previous setting.  Same results.
 *   dm_debug_strip_slab_budget   caps the bytes of slabs one channel group of the calling thread's
 *                                strip-path projections may use (0 = no cap; returns the previous
 *                                cap), so that value maps go through several channel groups -- the
 *                                route a many-class object map takes when its slabs exceed the
 *                                workspace -- at sizes the oracle finishes in seconds.
 *   dm_debug_last_strip_info     what the calling thread's last strip-path projection launched (its
 *                                value-map pass; a height pass behind it counts on top): out4 = {index-pass
 *                                launches (value maps of many channels: the pixels' cells once for
 *                                all channels), scatter / value-pass launches, combine launches,
 *                                channel groups}.
 *   dm_debug_force_fused_split   dm_orth_project_fused_f32 on the strip path: strips of this many
 *                                pixels and groups of this many frames per workgroup (1, 2, 4, 8)
 *                                whatever the cost model says (0 = back to the model).
 *   dm_debug_last_fused_split    what the calling thread's last dm_orth_project_fused_f32 took:
 *                                out4 = {strip width, strips, frames per group, groups}; zeros when
 *                                it did not run on the strip path.
 *   dm_debug_strip_geometry      host only (no GPU needed): the strip path's geometry for `p` and
 *                                the given frames exactly as the kernels derive it.
 *                                out_geom (B, 8 + 4*8) int32 per frame: {ok | inside << 8 (bit s of
 *                                inside: strip s's window was not clipped by the map), strips P, strip
 *                                width, slack cells, union window x0, z0, w, h, then 8 strip
 *                                windows {x0, z0, w, h}}; out_covers (B, mh, P, 2) uint32 or NULL:
 *                                per map row and strip the cells [lo, hi) the strip can reach
 *                                and the sub-span only it can reach (written straight to the
 *                                map), packed lo | hi << 16 (0 = none); out_bound[5] or NULL: {slack,
 *                                fits LDS, window cells, union rows, union cells} the launch
 *                                is sized with (valid for every yaw / position of the camera).
 *                                Returns P, 0 when the strip path does not apply to `p`.
 *   dm_debug_strip_geometry_dev  the same windows / edges from the device's own evaluation
 *                                (frames on the host and on the device; geom_dev: B * (336 + 48) bytes of 8-byte
 *                                aligned device scratch, copied back by
 *                                the caller); returns P, 0 (not applicable) or < 0.
 */
int dm_debug_last_path(void);
int dm_debug_last_flow_fused(void);      /* 1: the last dm_orth_project_flow_f32 of the calling thread computed
                                            the flow inside the projection kernel */
int dm_debug_flow_fused(int on);         /* non-zero: dm_orth_project_flow_f32 of the calling thread fuses where it
                                            can (default 0: two kernels); returns the previous setting */
int dm_debug_force_legacy_window(int on);
int dm_debug_force_strips(int strips);
int dm_debug_fill_split(int head_share);
int dm_debug_force_nt_fill(int mode);
int dm_debug_strip_value_list(int on);
size_t dm_debug_strip_slab_budget(size_t bytes);
void dm_debug_last_strip_info(int32_t* out4);
void dm_debug_force_fused_split(int strip_width, int frames_per_group);
void dm_debug_last_fused_split(int32_t* out4);
int dm_debug_strip_geometry(const dm_params* p, const dm_frame* frames, int32_t* out_geom,
                            uint32_t* out_covers, int32_t* out_bound);
int dm_debug_strip_geometry_dev(const dm_params* p, const dm_frame* frames_host,
                                const float* frames_dev, void* geom_dev, size_t geom_bytes,
                                void* stream);

/*
 * Test hook (GPU): the silent-drop bound of the LDS-windowed paths checked on the device.  Every
 * pixel is projected with the device's float32 arithmetic; counts_dev[3] (uint64) receives
 * {pixels landing in the map, those outside the window of the image part that owns them, those
 * outside their strip's per-row cover}.  windows_dev (B, pc*pr, 4) int32 {x0, z0, w, h} as
 * dm_debug_windows (pd = 1) or dm_debug_strip_geometry return them, covers_dev (B, mh, pc, 2)
 * uint32 as dm_debug_strip_geometry returns them, or NULL.  workspace_dev: B * 128 bytes.
 */
int dm_debug_count_escapes(const dm_params* p, const dm_frame* frames, const float* depth_dev,
                           const uint8_t* valid_dev, const int32_t* windows_dev, int32_t pc,
                           int32_t pr, int32_t wp, int32_t hp, const uint32_t* covers_dev,
                           unsigned long long* counts_dev, void* workspace_dev, void* stream);

/*
 * Test hook: caps the bytes of LDS-window slabs one channel group of the calling thread's
 * projections may use (0 = no cap; returns the previous cap), so that value maps of few
 * frames go through several channel groups -- the route a 40-class object map of a full
 * batch takes -- at sizes the oracle finishes in seconds.
 */
size_t dm_debug_slab_budget(size_t bytes);

/*
 * Measurement hook (bench.py): the next dm_orth_project_f32 call on this thread
 * records `event` (a hipEvent_t) on its stream right after the kernels that
 * produce out/mask and before the optional batch fuse, then forgets it.
 */
/* The same for the start of that sequence: the event is recorded right before the first
 * operation the call enqueues (after its host-side geometry). */
void dm_debug_record_before_projection(void* event);
void dm_debug_record_after_projection(void* event);

/*
 * dm_debug_planes: the strip path's compact planes for the groups several strips share (calls of at most four
 * strips and two output channels; k_strip_combine_planes behind the scatter kernel): -1 the default (on), 0 the
 * slabs + lists + k_strip_combine_one of rounds 2-4, 1 on.  Returns the previous setting.
 */
int dm_debug_planes(int mode);

#ifdef __cplusplus
}
#endif
#endif /* DUNGEON_MAPS_AMD_DEBUG_H */
