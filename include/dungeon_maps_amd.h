/*
 * dungeon_maps_amd.h -- C ABI of the MI355X-native depth -> top-down projector.
 *
 * The reference (Ending2015a/dungeon_maps) has no FFI of its own: its hot path
 * is the Python function dungeon_maps.maps.orth_project and the torch_scatter
 * call underneath it.  This header is the boundary a binding for that path
 * binds; each entry point cites the reference interface it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer named *_dev is DEVICE memory
 *    (hipMalloc / PyTorch caching allocator), contiguous, float32 unless said
 *    otherwise.  `frames` is a HOST array (pageable is fine): the call copies
 *    what it needs out of it and it may be reused as soon as the call returns.
 *    Scratch is a caller-provided workspace (dm_*_workspace_bytes, 256-byte
 *    aligned); the library allocates no device memory of its own and never
 *    frees caller memory.
 *  - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL =
 *    the default stream).  No host synchronisation inside.
 *  - return 0 on success, a negative dm_status otherwise; dm_last_error()
 *    returns a thread-local message.  No exceptions cross the ABI.
 *  - re-entrant; no global state besides thread-local data (error string,
 *    the remembered split of recent call shapes).  No signal handlers, no
 *    threads of its own.
 */
#ifndef DUNGEON_MAPS_AMD_H
#define DUNGEON_MAPS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM_ABI_VERSION 10

typedef enum dm_status {
  DM_OK = 0,
  DM_ERR_INVALID_ARGUMENT = -1,
  DM_ERR_UNSUPPORTED = -2,
  DM_ERR_WORKSPACE_TOO_SMALL = -3,
  DM_ERR_LAUNCH = -4,
  DM_ERR_PLAN_MISMATCH = -5
} dm_status;

/*
 * Status word: the projection entry points take `status_dev`, an int32 in memory the device can
 * write and the caller can read without synchronising (pinned host memory; device memory works
 * too), or NULL.  The kernels store dm_status_bits into it when they refuse work the host could
 * not refuse up front -- today: a frame of a PREPARED batch (dm_orth_project_prepared_f32) whose
 * pose record in the caller's device buffer no longer fits the plan it is projected with; such a
 * frame's maps come out holding the fill value.  The word is sticky (never cleared by the
 * library): a caller polls it before its next call and treats non-zero as a failed projection.
 * Every bit lives in a byte of its own and is raised by a byte store, so that kernels raising
 * different bits never overwrite each other's.
 */
typedef enum dm_status_bits {
  DM_STATUS_FRAME_DID_NOT_FIT = 0x1,
  DM_STATUS_LIST_OVERFLOW = 0x100
} dm_status_bits;

/* utils.Reduction (dungeon_maps/utils.py:52-76) */
typedef enum dm_reduction {
  DM_REDUCE_MAX = 0,
  DM_REDUCE_MIN = 1,
  DM_REDUCE_SUM = 2,
  DM_REDUCE_MEAN = 3,
  DM_REDUCE_PROD = 4
} dm_reduction;

/*
 * Per-frame camera state, one 128-byte record per frame b (host array).
 * Built on the host with the reference's float32 op order so that no device
 * sin/cos is involved:
 *   Rp  = rotate([1,0,0], cam_pitch[b])   utils.py:303-327 via maps.py:790
 *   Ry  = rotate([0,1,0], cam_pose[b,2])  utils.py:303-327 via maps.py:884-885
 * R is stored row-major exactly as utils.py:326-327 builds it; a point is
 * transformed as out_i = sum_j R[3*j+i] * p_j (utils.py:329).
 */
typedef struct dm_frame {
  float Rp[9];
  float cam_height;      /* maps.py:791-797 */
  float Ry[9];
  float tx, tz;          /* cam_pose[b,0], cam_pose[b,1]  (maps.py:887-892) */
  float width_offset;    /* maps.py:1004 */
  float height_offset;   /* maps.py:1005 */
  float reserved[9];     /* dm_camera_affine_grid_f32: rotate([1,0,0], -cam_pitch[b])
                            (maps.py:845); unused by the projector */
} dm_frame;

/*
 * Host-only helper: the per-call columns of a dm_frame table.  `static_table` (B, 32 floats)
 * holds what does not change from call to call for one camera rig (pitch rotation, camera
 * height, offsets, a neutral yaw); `pose` (B, 3) is cam_pose = (x, z, yaw); `sin_yaw` / `cos_yaw`
 * (B) are sin / cos of the yaw column as the CALLER's libm gives them (the Python side passes
 * torch's CPU results -- the values the reference itself computes with, utils.py:318-327 --
 * so that no second libm is involved).  out_table = static_table with, per frame,
 *   Ry = rotate([0,1,0], yaw) built element-wise as the reference builds it
 *        (R = (I + sin S) + (1 - cos) S^2; |yaw| <= 1e-3 clamps to the identity, utils.py:323-324),
 *   tx, tz = pose x, z  (maps.py:887-892).
 * No GPU involved; returns DM_OK or DM_ERR_INVALID_ARGUMENT.
 */
int dm_frames_fill_f32(const float* static_table, int32_t B, const float* pose, const float* sin_yaw,
                       const float* cos_yaw, float* out_table);

/*
 * Call-wide parameters of orth_project (maps.py:127-153).
 *   depth   (B, dc, H, W)      value (B, vc, H, W) or NULL (vc = 0: project
 *   heights, maps.py:311-313)  valid (B, valid_c, H, W) uint8/bool or NULL
 * dc == 1 broadcasts the cell index over the vc value channels (torch_scatter
 * index broadcast, utils.py:475-477); otherwise dc must equal vc.
 */
typedef struct dm_params {
  int32_t B, dc, vc, H, W;
  int32_t mh, mw;            /* map_height, map_width */
  int32_t clip_border;       /* <=0: off            maps.py:273-277 */
  int32_t flip_h;            /* maps.py:670-671, 1006-1009 */
  int32_t to_global;         /* maps.py:290-295 */
  int32_t reduction;         /* dm_reduction        utils.py:470-477 */
  int32_t has_dmin, has_dmax, has_hmax;  /* trunc_* is not None */
  int32_t valid_c;           /* 0 (no valid map), 1 or dc */
  float cx, cy, fx, fy;      /* utils.py:94-116 cast to f32 (maps.py:672-675) */
  float res;                 /* map_res */
  float fill;                /* fill_value; None => 0 (zeros canvas, maps.py:320) */
  float dmin, dmax, hmax;    /* maps.py:537-544, 286-288 */
} dm_params;

int dm_version(void);
const char* dm_last_error(void);

/* Scratch bytes dm_orth_project_f32 / dm_orth_project_fused_f32 need for `p`. */
size_t dm_orth_project_workspace_bytes(const dm_params* p);

/*
 * Replaces orth_project (maps.py:127-351): depth_map_to_point_cloud (462-545),
 * _mask_borders (48-70), camera_to_local_space (753-800), height truncation
 * (286-288), local_to_global_space (850-895), map_quantize (944-1019),
 * project (1089-1173) and utils.scatter_tensor incl. the torch_scatter call
 * (utils.py:389-492), fused.
 *   out_dev    (B, oc, mh, mw) f32, oc = vc ? vc : dc      -- topdown_map
 *   mask_dev   (B, oc, mh, mw) u8 (0/1)                    -- masks
 *   height_dev (B, dc, mh, mw) f32 or NULL: the second, NINF/max projection
 *              of maps.py:332-350 (only meaningful when vc > 0; with vc == 0
 *              the height map IS out, maps.py:333-334)
 *   fused_dev / fused_mask_dev (oc, mh, mw) f32 / u8, or both NULL: the maps
 *              of the B frames fused into one (max or min over the batch axis
 *              -- MapBuilder.merge / fuse_topdown_maps, maps.py:2181-2287,
 *              2471-2508, for maps sharing one frame) and its mask.
 *              The per-rank partial of the "projected+fused" metric.
 * Outputs are fully written (no pre-initialisation needed).  On the strip path (max / min with
 * both depth truncations, the projector's axis-aligned rotations, one pitch per batch) the call
 * enqueues kernels only: the frames' camera state travels in the scatter kernel's arguments
 * (up to 64 frames per launch), nothing is copied to the device.
 */
int dm_orth_project_f32(const dm_params* p, const dm_frame* frames,
                        const float* depth_dev, const float* value_dev,
                        const uint8_t* valid_dev, float* out_dev,
                        uint8_t* mask_dev, float* height_dev,
                        float* fused_dev, uint8_t* fused_mask_dev,
                        void* workspace_dev, size_t workspace_bytes,
                        int32_t* status_dev, void* stream);

/*
 * Batch-fused projection: all B frames are reduced (max or min only) into ONE
 * (oc, mh, mw) map in the shared global frame -- the per-rank partial of the
 * "projected+fused" metric.  Equals the reference's MapBuilder.merge /
 * fuse_topdown_maps (maps.py:2181-2287, 2471-2508) for maps that share
 * res/offsets/size, which is an element-wise max.
 *   out_dev (oc, mh, mw) f32, mask_dev (oc, mh, mw) u8
 * If `accumulate` is non-zero, out_dev's current content takes part in the
 * reduction (running world map); otherwise it starts from p->fill.
 */
int dm_orth_project_fused_f32(const dm_params* p, const dm_frame* frames,
                              const float* depth_dev, const float* value_dev,
                              const uint8_t* valid_dev, float* out_dev,
                              uint8_t* mask_dev, int accumulate,
                              void* workspace_dev, size_t workspace_bytes,
                              int32_t* status_dev, void* stream);

/*
 * orth_project and camera_affine_grid of the SAME depth maps in one call (the reference's ego-flow
 * demo runs both on every frame, demos/ego_flow/run.py:75-90; BASELINE configs[4]): the projection
 * of dm_orth_project_f32 (heights: p->vc = 0) plus the flow grid of dm_camera_affine_grid_f32,
 * `flow_frames` being that entry point's frame table (the pose TRANSITION in Ry / tx / tz,
 * rotate(X, -pitch) in reserved) and grid_dev (B, dc, H, W, 2) f32.  Where the LDS-window path's
 * lean height kernel takes the projection (both depth bounds finite, no valid map / border /
 * height truncation, max or min, no depth bands) that kernel CAN compute the flow of every pixel
 * from the depth it has loaded -- one depth read for both results (dm_debug_flow_fused(1)); measured
 * on MI355X the two kernels one after the other are the faster form (the flow's divisions want the
 * stand-alone kernel's occupancy), so that is what runs by default.  Bit-identical results either
 * way (dm_debug_last_flow_fused tells which ran).  Workspace: dm_orth_project_workspace_bytes(p).
 */
int dm_orth_project_flow_f32(const dm_params* p, const dm_frame* frames, const dm_frame* flow_frames,
                             const float* depth_dev, const uint8_t* valid_dev, float* out_dev,
                             uint8_t* mask_dev, float* fused_dev, uint8_t* fused_mask_dev,
                             float* grid_dev, void* workspace_dev, size_t workspace_bytes,
                             int32_t* status_dev, void* stream);

/*
 * camera_affine_grid (maps.py:353-460): per pixel of depth (B, dc, H, W) the
 * (column, row) it maps to after the camera moved by trans_pose -- frames[b]
 * holds Rp = rotate(X, pitch), cam_height, Ry/tx/tz = the pose TRANSITION
 * (maps.py:435-439) and reserved = rotate(X, -pitch).  Uses p->B, dc, H, W,
 * cx, cy, fx, fy, flip_h.  grid_dev (B, dc, H, W, 2) f32.  The workspace needs
 * B * sizeof(dm_frame) bytes.
 */
int dm_camera_affine_grid_f32(const dm_params* p, const dm_frame* frames,
                              const float* depth_dev, float* grid_dev,
                              void* workspace_dev, size_t workspace_bytes, void* stream);

/*
 * fuse_topdown_maps (maps.py:2039-2287) without materialised point clouds.  Every valid
 * cell of a source map becomes a point at its cell centre (map_dequantize, maps.py:1021-1087:
 * x = (col - woff) * res, z = ((flip ? (mh-1) - row : row) - hoff) * res, y = height), goes
 * local -> global with the map's own pose (rotate, then translate; skipped if the map is
 * already global) and global -> local with the target's (translate, then rotate; skipped
 * for a global target), and is quantised in the target frame (map_quantize).
 *   dm_fuse_bbox_f32     quantises with zero offsets, unflipped, and reduces the bounding
 *                        box of the valid cells into stats_dev[5] = {min col, max col,
 *                        min row, max row, any valid} (int32; init != 0 resets it first)
 *                        -- the reference's two .item() syncs become one copy of 5 ints.
 *   dm_fuse_scatter_f32  quantises with the final offsets / flip / size and reduces
 *                        (max or min) the cell's value (value_dev, or its height) into
 *                        canvas_dev (b, c, mh, mw) and, if height_canvas_dev is given, its
 *                        height (max) into that (b, c, mh, mw) as well.  The canvases must hold
 *                        their fill values; masks follow with dm_mask_from_map_f32.
 *   dm_fuse_bbox_multi_f32 / dm_fuse_scatter_multi_f32   the same for 1 .. DM_FUSE_MAX_SOURCES
 *                        source maps (equal b and c) in ONE launch each -- MapBuilder.merge
 *                        (maps.py:2357-2406) fuses two maps per frame.  The bounding box comes
 *                        back as five MAXIMA of unsigned words, so that a zero-filled
 *                        stats_dev[5] needs no initialising launch: with u(x) = (uint32)x ^
 *                        0x80000000, stats = {max ~u(col), max u(col), max ~u(row), max u(row),
 *                        any valid}; stats_dev must hold zeros on entry.
 * Rotations are row-major 3x3 as utils.py:326-327, applied as the reference's FMA chain.
 */
#define DM_FUSE_MAX_BATCH 8
#define DM_FUSE_MAX_SOURCES 4
typedef struct dm_fuse_src {
  const float* height_dev;    /* (b, hc, h, w) cell heights */
  const uint8_t* mask_dev;    /* (b, mc, h, w) valid cells */
  const float* value_dev;     /* (b, c, h, w) or NULL: the heights are the values */
  int32_t b, c, hc, mc;       /* hc, mc in {1, c} */
  int32_t h, w;
  int32_t flip_h;             /* source map's flip_h */
  int32_t has_l2g, has_g2l;
  float res;                  /* source map_res */
  float target_res;           /* target map_res */
  float woff[DM_FUSE_MAX_BATCH], hoff[DM_FUSE_MAX_BATCH];       /* source offsets per batch row */
  float l2g[DM_FUSE_MAX_BATCH][12];   /* R (9) | t (3): out = rotate(p) + t */
  float g2l[DM_FUSE_MAX_BATCH][12];   /* R (9) | t (3): out = rotate(p + t) */
} dm_fuse_src;

int dm_fuse_bbox_f32(const dm_fuse_src* src, int32_t* stats_dev, int init, void* stream);
int dm_fuse_scatter_f32(const dm_fuse_src* src, float width_offset, float height_offset,
                        int flip_h, int64_t map_height, int64_t map_width, int reduction,
                        float* canvas_dev, float* height_canvas_dev, void* stream);
int dm_fuse_bbox_multi_f32(const dm_fuse_src* srcs, int32_t n, int32_t* stats_dev, void* stream);
/* The five words of a bounding box on the host: enqueues the copy on `stream` (through a pinned block the
 * library keeps per host thread) and WAITS for the stream -- the one host synchronisation of fuse_topdown_maps
 * (the reference has two .item() calls there, maps.py:2146-2179).  ABI v10. */
int dm_fuse_bbox_read_i32(const int32_t* stats_dev, int32_t* stats_host, void* stream);
int dm_fuse_scatter_multi_f32(const dm_fuse_src* srcs, int32_t n, float width_offset,
                              float height_offset, int flip_h, int64_t map_height,
                              int64_t map_width, int reduction, float* canvas_dev,
                              float* height_canvas_dev, void* stream);

/*
 * crop_topdown_map / TopdownMap.select (maps.py:1959-2037): generate_crop_grid
 * (utils.py:571-611) + image_sample(mode='nearest') (utils.py:613-652) fused into one
 * gather.  image_dev (B, C, h, w) f32; mask_dev (B, C, h, w) uint8/bool or NULL (the
 * companion mask rides on the same coordinates with fill False); center_dev (B, 2) f32
 * crop centres (x, y) in pixels; out_dev (B, C, crop_h, crop_w), out_mask_dev likewise.
 * has_fill == 0 is fill_value=None: zero padding and padding_mode='zeros' (utils.py:634-637);
 * otherwise the image is padded with `fill` and coordinates are clamped ('border').
 * Pixel centres follow grid_sample(align_corners=True), nearest = round half to even.
 */
int dm_crop_nearest_f32(const float* image_dev, const uint8_t* mask_dev, const float* center_dev,
                        int64_t B, int64_t C, int64_t h, int64_t w, int64_t crop_h, int64_t crop_w,
                        float fill, int has_fill, float* out_dev, uint8_t* out_mask_dev,
                        void* stream);

/*
 * The same crop with any of image_sample's modes (utils.py:613-652, maps.py:1959-2037): `mode` nearest (the
 * gather above), bilinear or bicubic as torch's grid_sample(align_corners=True) evaluates them on the image padded
 * by one pixel per side ('border' with the pad = fill, 'zeros' for has_fill == 0) -- float32, one rounding per
 * operation; within a few ulp of torch's CPU kernel (which sums in another order), the same NaN / inf where an
 * empty (-inf) cell meets a zero weight.  The companion mask (fill False) is sampled as 0 / 1 and written as
 * "nonzero", as the reference does.
 */
typedef enum dm_sample_mode { DM_SAMPLE_NEAREST = 0, DM_SAMPLE_BILINEAR = 1, DM_SAMPLE_BICUBIC = 2 } dm_sample_mode;
int dm_crop_sample_f32(const float* image_dev, const uint8_t* mask_dev, const float* center_dev,
                       int64_t B, int64_t C, int64_t h, int64_t w, int64_t crop_h, int64_t crop_w,
                       float fill, int has_fill, int mode, float* out_dev, uint8_t* out_mask_dev,
                       void* stream);

/*
 * utils.rotate + utils.translate on materialised points (utils.py:229-330; used
 * by camera_to_local_space / local_to_global_space / global_to_local_space /
 * local_to_camera_space, maps.py:753-942).  pts_dev, out_dev (B, n, 3) f32;
 * R_dev (B, 9) row-major as utils.py:326-327; t_dev (B, 3).
 *   translate_first == 0:  out = rotate(p) + t        (camera->local, local->global)
 *   translate_first != 0:  out = rotate(p + t)        (global->local, local->camera)
 * rotate is out_i = fma(p2,R[6+i], fma(p1,R[3+i], p0*R[i])) -- the reference's
 * CPU bmm order.  In-place (out_dev == pts_dev) is allowed.
 */
int dm_affine_points_f32(const float* pts_dev, const float* R_dev, const float* t_dev,
                         int64_t B, size_t n, int translate_first, float* out_dev,
                         void* stream);

/*
 * map_quantize (maps.py:944-1019): xb = floor(x/res + woff[b] + 0.5),
 * zb = floor((flip ? (mh-1) - (z/res + hoff[b]) : z/res + hoff[b]) + 0.5) as
 * int64 (NaN / out of range -> INT64_MIN, like the reference on x86).
 * x_dev, z_dev (B, n) f32; woff_dev, hoff_dev (B) f32; xb_dev, zb_dev (B, n) i64.
 */
int dm_map_quantize_f32(const float* x_dev, const float* z_dev, const float* woff_dev,
                        const float* hoff_dev, int64_t B, size_t n, float res,
                        int32_t map_height, int32_t flip_h, int64_t* xb_dev,
                        int64_t* zb_dev, void* stream);

/*
 * utils.scatter_tensor incl. the torch_scatter call (utils.py:389-492) on
 * pre-ravelled indices: values_dev (R, C, N) f32; index_dev (R, Ci, N) i64 with
 * Ci in {1, C}, a flat cell in [0, M) or anything else = dropped;
 * canvas_dev (R, C, M) f32 in/out -- filled with `fill` first if has_fill, else
 * its content takes part; mask_dev (R, C, M) u8 = cell changed (utils.py:489-491).
 */
size_t dm_scatter_workspace_bytes(int64_t R, int32_t C, size_t M, int32_t has_fill,
                                  int32_t reduction);
int dm_scatter_f32(const float* values_dev, const int64_t* index_dev, float* canvas_dev,
                   uint8_t* mask_dev, int64_t R, int32_t C, int32_t Ci, size_t N, size_t M,
                   float fill, int32_t has_fill, int32_t reduction, void* workspace_dev,
                   size_t workspace_bytes, void* stream);

/*
 * Prepared frames: the camera state of a batch kept in a device buffer of the caller's, so that a
 * captured launch sequence (HIP graph) can project OTHER poses when replayed -- the kernels of
 * dm_orth_project_prepared_f32 read the frames' pose records (48 bytes each) from `prepared_dev`
 * instead of from their own arguments; everything else is dm_orth_project_f32's strip path.  To
 * project other poses through a captured graph, call dm_frames_prepare_f32 again on the same buffer
 * (stream ordered) with `must_match` = the captured plan.
 *
 * Applies where the strip path does (max / min, trunc_depth_min >= 0 and a finite
 * trunc_depth_max, map_width and image width multiples of 4, one camera pitch per batch, and the
 * windows of all strips fit in LDS for every yaw); otherwise dm_frames_prepare_f32 returns
 * DM_ERR_UNSUPPORTED and the caller stays with dm_orth_project_f32.
 *
 *   dm_frames_prepared_bytes     device bytes a prepared batch of p->B frames needs (0: the strip
 *                                path never applies to `p`); the buffer must be 256-byte aligned.
 *   dm_frames_prepare_f32        validates `frames` (host, as for dm_orth_project_f32) and derives the
 *                                launch plan ON THE HOST first; if `must_match` is given and the plan
 *                                differs from it, returns DM_ERR_PLAN_MISMATCH and enqueues NOTHING
 *                                (`prepared_dev` keeps the records it held).  Otherwise fills
 *                                `plan_out` and enqueues ONE copy of p->B pose records into
 *                                `prepared_dev`.  What it replaces in the reference: the per-call pose
 *                                handling of MapProjector.orth_project (maps.py:1406-1465: cam_pose ->
 *                                rotation and translation tensors, utils.py:229-330) hoisted out of
 *                                the per-batch call.
 *   dm_orth_project_prepared_f32 dm_orth_project_f32 with the frames taken from `prepared_dev`; `plan`
 *                                must be the one dm_frames_prepare_f32 returned for these parameters
 *                                (checked: strips, strip width, LDS window / table sizes, slack), else
 *                                DM_ERR_UNSUPPORTED.  A frame whose record in `prepared_dev` does not
 *                                fit `plan` (the buffer changed behind the plan's back) projects
 *                                nothing and raises DM_STATUS_FRAME_DID_NOT_FIT in `status_dev`.
 *                                One projection at a time per workspace (stream ordered).
 */
typedef struct dm_frames_plan {
  int32_t strips;            /* column strips per frame */
  int32_t strip_width;       /* pixels */
  int32_t slab_cells;        /* cells of the largest map window a strip can have, any yaw */
  int32_t max_rows;          /* rows / cells of the largest union window of a frame */
  int32_t max_union_cells;
  int32_t slack_cells;       /* float32 slack the windows carry */
  int32_t magnitude;         /* bound (cells, quantised) on the frames' offsets / translations the slack was derived for */
  float pitch[4];            /* the batch's pitch rotation: Rp[4], Rp[5], Rp[7], Rp[8] */
  int32_t reserved;
} dm_frames_plan;

size_t dm_frames_prepared_bytes(const dm_params* p);
int dm_frames_prepare_f32(const dm_params* p, const dm_frame* frames, void* prepared_dev,
                          size_t prepared_bytes, const dm_frames_plan* must_match,
                          dm_frames_plan* plan_out, void* stream);
int dm_orth_project_prepared_f32(const dm_params* p, const dm_frames_plan* plan, const void* prepared_dev,
                                 const float* depth_dev, const float* value_dev,
                                 const uint8_t* valid_dev, float* out_dev, uint8_t* mask_dev,
                                 float* height_dev, float* fused_dev, uint8_t* fused_mask_dev,
                                 void* workspace_dev, size_t workspace_bytes, int32_t* status_dev,
                                 void* stream);

/*
 * Which measurement / instrumentation flags the library was BUILT with, as a string ("" for the product
 * build: tests/test_codegen_guard.py requires that of the shipped library; "DM_STAMPS", "DM_HOST_TIMING"
 * for the instrumented builds of tools/).  Builds that change results for timing's sake do not exist in
 * this tree (tools/experiments keeps them as patches) and are refused by the headers (#error).
 */
const char* dm_build_flags(void);

/* Test / measurement hooks (dm_debug_*): include/dungeon_maps_amd_debug.h. */


/*
 * Fuse B maps that share one frame (same res/offsets/size) into one:
 * out[i] = reduce_b maps[b, i], reduce = max or min.  This is what
 * MapBuilder.merge -> fuse_topdown_maps (maps.py:2181-2287, 2471-2508) computes
 * for such maps (element-wise max in a shared frame), applied
 * across the batch axis.  maps_dev (B, n) f32, out_dev (n) f32; if
 * `accumulate` is non-zero out_dev's current content takes part.
 */
int dm_fuse_batch_f32(const float* maps_dev, int64_t B, size_t n, float* out_dev,
                      int reduction, int accumulate, void* stream);

/*
 * mask = (map - fill != 0) with NaN -> 0: utils.py:489-491 as a function of
 * the finished map (used after the cross-rank max all-reduce).  n elements.
 */
int dm_mask_from_map_f32(const float* map_dev, float fill, uint8_t* mask_dev,
                         size_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DUNGEON_MAPS_AMD_H */
