/*
 * dm_oracle.c -- CPU restatement of the reference's orth_project hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or
 * call it.  Nothing under dungeon_maps_amd/ imports it; the product path is
 * the HIP library and fails loudly when that is missing.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this code against
 * tests/golden/*.npz, which were produced by running the reference's own
 * dungeon_maps/maps.py + utils.py in the build container
 * (tests/golden/gen_golden.py; torch-scatter, an absent third-party
 * dependency, restated there).  Integer bins and validity are bit-exact on
 * every pixel; max/min maps are bit-exact; sum/mean/prod within tolerance.
 *
 * Every step cites the reference line it restates (paths relative to
 * /root/reference/dungeon_maps/).  All arithmetic is float32, one rounding per
 * written operation (build with -ffp-contract=off); the only fused operations
 * are the explicit fmaf() of the 3-term rotation dot product, which is how the
 * reference's einsum->bmm evaluates on CPU (utils.py:329).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

enum { DMO_MAX = 0, DMO_MIN = 1, DMO_SUM = 2, DMO_MEAN = 3, DMO_PROD = 4 };

typedef struct {
  int32_t B;           /* frames */
  int32_t dc;          /* depth channels (index channels), maps.py:234 */
  int32_t vc;          /* value channels; 0 => project heights (maps.py:311-313) */
  int32_t H, W;        /* depth image size */
  int32_t mh, mw;      /* map size */
  int32_t clip_border; /* maps.py:273-277 */
  int32_t flip_h;      /* maps.py:670-671, 1006-1009 */
  int32_t to_global;   /* maps.py:290-295 */
  int32_t reduction;   /* utils.py:52-76 */
  int32_t has_dmin, has_dmax, has_hmax; /* None => disabled */
  int32_t valid_c;     /* channels of valid map (0 = none, 1 or dc) */
  float cx, cy, fx, fy; /* utils.py:94-116, cast to f32 at maps.py:672-675 */
  float res;           /* map_res as f32 (python float / f32 tensor, maps.py:1004) */
  float fill;          /* fill_value (None => 0: zeros canvas, maps.py:320) */
  float dmin, dmax, hmax;
} dmo_params;

/* one pixel: maps.py:462-545 (unproject), 753-800 (pitch+height),
 * 850-895 (yaw+translate), 944-1019 (quantize), 1150-1158 + utils.py:447-456
 * (bounds).  Returns 1 when the point lands in the canvas. */
static inline int dmo_pixel(const dmo_params* p, float z, int r, int q,
                            int valid_in, const float* Rp, float cam_h,
                            const float* Ry, float tx, float tz, float woff,
                            float hoff, float* y_out, int64_t* xb_out,
                            int64_t* zb_out, int* valid_pre) {
  /* generate_image_coords utils.py:562-563; flip maps.py:670-671 */
  float xq = (float)q;
  float yr = (float)r;
  if (p->flip_h) yr = (float)(p->H - 1) - yr;
  /* maps.py:677-678: (x-cx)/fx * z -- sub, true division, mul */
  float X = ((xq - p->cx) / p->fx) * z;
  float Y = ((yr - p->cy) / p->fy) * z;
  float Z = z;
  /* maps.py:537-544 */
  int valid = 1;
  if (p->has_dmax) valid = valid && (z <= p->dmax);
  if (p->has_dmin) valid = valid && (z >= p->dmin);
  valid = valid && valid_in;
  /* maps.py:48-70 */
  if (p->clip_border > 0) {
    int c = p->clip_border;
    if (r < c || r >= p->H - c || q < c || q >= p->W - c) valid = 0;
  }
  /* utils.py:329: out_i = sum_j R[j][i] * p_j, bmm accumulates j = 0,1,2 with
   * FMA; utils.py:256 translate adds (0, cam_height, 0) */
  float x1 = fmaf(Z, Rp[6], fmaf(Y, Rp[3], X * Rp[0])) + 0.0f;
  float y1 = fmaf(Z, Rp[7], fmaf(Y, Rp[4], X * Rp[1])) + cam_h;
  float z1 = fmaf(Z, Rp[8], fmaf(Y, Rp[5], X * Rp[2])) + 0.0f;
  /* maps.py:286-288 */
  if (p->has_hmax) valid = valid && (y1 <= p->hmax);
  float x2 = x1, y2 = y1, z2 = z1;
  if (p->to_global) {
    /* maps.py:885-892 */
    x2 = fmaf(z1, Ry[6], fmaf(y1, Ry[3], x1 * Ry[0])) + tx;
    y2 = fmaf(z1, Ry[7], fmaf(y1, Ry[4], x1 * Ry[1])) + 0.0f;
    z2 = fmaf(z1, Ry[8], fmaf(y1, Ry[5], x1 * Ry[2])) + tz;
  }
  /* maps.py:1004-1013 */
  float xf = x2 / p->res + woff;
  float zf = z2 / p->res + hoff;
  if (p->flip_h) zf = (float)(p->mh - 1) - zf;
  xf = floorf(xf + 0.5f);
  zf = floorf(zf + 0.5f);
  /* .to(int64) of NaN / out-of-range floats is INT64_MIN on x86 */
  int64_t xb = (xf >= -9.2233720368547758e18f && xf < 9.2233720368547758e18f)
                   ? (int64_t)xf : INT64_MIN;
  int64_t zb = (zf >= -9.2233720368547758e18f && zf < 9.2233720368547758e18f)
                   ? (int64_t)zf : INT64_MIN;
  if (valid_pre) *valid_pre = valid;
  /* maps.py:1150-1158 */
  valid = valid && xb >= 0 && xb < p->mw && zb >= 0 && zb < p->mh;
  *y_out = y2;
  *xb_out = xb;
  *zb_out = zb;
  return valid;
}

static inline void dmo_reduce(int red, float* cell, float v) {
  switch (red) {
    case DMO_MAX: if (v > *cell) *cell = v; break;   /* torch-scatter: new > old */
    case DMO_MIN: if (v < *cell) *cell = v; break;
    case DMO_SUM: case DMO_MEAN: *cell = *cell + v; break;
    case DMO_PROD: *cell = *cell * v; break;
  }
}

/* utils.py:489-491: mask = nan_to_num(|canvas - copy|, 0, 1, 1) != 0 */
static inline uint8_t dmo_mask(float out, float fill) {
  float d = fabsf(out - fill);
  if (isnan(d)) return 0;
  return d != 0.0f;
}

/*
 * depth  (B, dc, H, W) f32        value (B, vc, H, W) f32 or NULL
 * valid  (B, valid_c, H, W) u8 or NULL
 * Rp, Ry (B, 9) row-major R as utils.py:326-327; trans (B,2) = pose x, z
 * out    (B, oc, mh, mw) f32 with oc = vc ? vc : dc;   mask same shape u8
 * height (B, dc, mh, mw) f32 or NULL  (maps.py:332-350; broadcast over oc by
 *         the caller)
 * dbg_xb/dbg_zb (B, dc, H*W) i64, dbg_valid (B, dc, H*W) u8, dbg_y f32: NULL
 *         or per-pixel intermediates (valid = before the canvas bounds test,
 *         as the reference's flat_mask at maps.py:300)
 */
int dmo_orth_project(const dmo_params* p, const float* depth, const float* value,
                     const uint8_t* valid, const float* Rp, const float* cam_h,
                     const float* Ry, const float* trans, const float* woff,
                     const float* hoff, float* out, uint8_t* mask, float* height,
                     int64_t* dbg_xb, int64_t* dbg_zb, uint8_t* dbg_valid,
                     float* dbg_y, int nthreads) {
  const int B = p->B, dc = p->dc, vc = p->vc, H = p->H, W = p->W;
  const int64_t N = (int64_t)H * W, M = (int64_t)p->mh * p->mw;
  const int oc = vc ? vc : dc;
  if (vc && !(dc == 1 || dc == vc)) return -1;
  if (valid && !(p->valid_c == 1 || p->valid_c == dc)) return -2;
  const float ninf = -INFINITY;
  const int red = p->reduction;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int b = 0; b < B; ++b) {
    const float* rp = Rp + 9 * b;
    const float* ry = Ry ? Ry + 9 * b : rp;
    const float tx = trans ? trans[2 * b] : 0.0f, tz = trans ? trans[2 * b + 1] : 0.0f;
    float* ob = out + (int64_t)b * oc * M;
    /* utils.py:472-473 fill (None => canvas stays zeros) */
    for (int64_t i = 0; i < oc * M; ++i) ob[i] = p->fill;
    if (height) {
      float* hb = height + (int64_t)b * dc * M;
      for (int64_t i = 0; i < dc * M; ++i) hb[i] = ninf;
    }
    float* cnt = NULL;
    if (red == DMO_MEAN) cnt = (float*)calloc((size_t)(oc * M), sizeof(float));
    for (int ch = 0; ch < dc; ++ch) {
      const float* dimg = depth + ((int64_t)b * dc + ch) * N;
      const uint8_t* vimg = valid
          ? valid + ((int64_t)b * p->valid_c + (p->valid_c == 1 ? 0 : ch)) * N : NULL;
      for (int r = 0; r < H; ++r) {
        for (int q = 0; q < W; ++q) {
          const int64_t i = (int64_t)r * W + q;
          float y; int64_t xb, zb; int vpre;
          int ok = dmo_pixel(p, dimg[i], r, q, vimg ? (vimg[i] != 0) : 1, rp,
                             cam_h[b], ry, tx, tz, woff[b], hoff[b], &y, &xb, &zb,
                             &vpre);
          if (dbg_xb) {
            const int64_t o = ((int64_t)b * dc + ch) * N + i;
            dbg_xb[o] = xb; dbg_zb[o] = zb; dbg_valid[o] = (uint8_t)vpre; dbg_y[o] = y;
          }
          if (!ok) continue;
          const int64_t cell = zb * p->mw + xb;   /* ravel_index utils.py:332-370 */
          if (!vc) {
            dmo_reduce(red, ob + (int64_t)ch * M + cell, y);
            if (cnt) cnt[(int64_t)ch * M + cell] += 1.0f;
          } else if (dc == 1) {
            for (int k = 0; k < vc; ++k) {     /* index broadcast over channels */
              dmo_reduce(red, ob + (int64_t)k * M + cell,
                         value[((int64_t)b * vc + k) * N + i]);
              if (cnt) cnt[(int64_t)k * M + cell] += 1.0f;
            }
          } else {
            dmo_reduce(red, ob + (int64_t)ch * M + cell,
                       value[((int64_t)b * vc + ch) * N + i]);
            if (cnt) cnt[(int64_t)ch * M + cell] += 1.0f;
          }
          if (height) {                         /* maps.py:337-348: NINF, max */
            float* hc = height + ((int64_t)b * dc + ch) * M + cell;
            if (y > *hc) *hc = y;
          }
        }
      }
    }
    if (cnt) {   /* torch-scatter scatter_mean: sum into out, / clamp(count,1) */
      for (int64_t i = 0; i < oc * M; ++i) ob[i] = ob[i] / (cnt[i] < 1.0f ? 1.0f : cnt[i]);
      free(cnt);
    }
    uint8_t* mb = mask + (int64_t)b * oc * M;
    for (int64_t i = 0; i < oc * M; ++i) mb[i] = dmo_mask(ob[i], p->fill);
  }
  return 0;
}

/* Fused global map (north_star "projected+fused"): every frame of the batch is
 * reduced with max into ONE (oc, mh, mw) map that starts at fill.  Equivalent
 * to max over dim 0 of dmo_orth_project's output when all frames share
 * res/offsets/size (the reference's fuse_topdown_maps is an
 * element-wise max in a shared global frame). */
int dmo_orth_project_fused(const dmo_params* p, const float* depth,
                           const float* value, const uint8_t* valid,
                           const float* Rp, const float* cam_h, const float* Ry,
                           const float* trans, const float* woff,
                           const float* hoff, float* out, uint8_t* mask) {
  const int B = p->B, dc = p->dc, vc = p->vc, H = p->H, W = p->W;
  const int64_t N = (int64_t)H * W, M = (int64_t)p->mh * p->mw;
  const int oc = vc ? vc : dc;
  if (p->reduction != DMO_MAX && p->reduction != DMO_MIN) return -3;
  if (vc && !(dc == 1 || dc == vc)) return -1;
  for (int64_t i = 0; i < oc * M; ++i) out[i] = p->fill;
  for (int b = 0; b < B; ++b) {
    const float* rp = Rp + 9 * b;
    const float* ry = Ry ? Ry + 9 * b : rp;
    const float tx = trans ? trans[2 * b] : 0.0f, tz = trans ? trans[2 * b + 1] : 0.0f;
    for (int ch = 0; ch < dc; ++ch) {
      const float* dimg = depth + ((int64_t)b * dc + ch) * N;
      const uint8_t* vimg = valid
          ? valid + ((int64_t)b * p->valid_c + (p->valid_c == 1 ? 0 : ch)) * N : NULL;
      for (int r = 0; r < H; ++r)
        for (int q = 0; q < W; ++q) {
          const int64_t i = (int64_t)r * W + q;
          float y; int64_t xb, zb;
          if (!dmo_pixel(p, dimg[i], r, q, vimg ? (vimg[i] != 0) : 1, rp, cam_h[b],
                         ry, tx, tz, woff[b], hoff[b], &y, &xb, &zb, NULL))
            continue;
          const int64_t cell = zb * p->mw + xb;
          if (!vc) dmo_reduce(p->reduction, out + (int64_t)ch * M + cell, y);
          else if (dc == 1)
            for (int k = 0; k < vc; ++k)
              dmo_reduce(p->reduction, out + (int64_t)k * M + cell,
                         value[((int64_t)b * vc + k) * N + i]);
          else
            dmo_reduce(p->reduction, out + (int64_t)ch * M + cell,
                       value[((int64_t)b * vc + ch) * N + i]);
        }
    }
  }
  for (int64_t i = 0; i < oc * M; ++i) mask[i] = dmo_mask(out[i], p->fill);
  return 0;
}

int dmo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* utils.rotate + utils.translate on explicit points (utils.py:229-330), the
 * restatement behind camera_to_local_space / local_to_global_space /
 * global_to_local_space / local_to_camera_space (maps.py:753-942).
 * pts, out (B, n, 3); R (B, 9); t (B, 3).  translate_first: out = rot(p + t). */
void dmo_affine_points(const float* pts, const float* R, const float* t, int B, int64_t n,
                       int translate_first, float* out) {
  for (int b = 0; b < B; ++b) {
    const float* r = R + 9 * b;
    const float* tb = t + 3 * b;
    for (int64_t i = 0; i < n; ++i) {
      const float* p = pts + ((int64_t)b * n + i) * 3;
      float p0 = p[0], p1 = p[1], p2 = p[2];
      if (translate_first) { p0 = p0 + tb[0]; p1 = p1 + tb[1]; p2 = p2 + tb[2]; }
      float o0 = fmaf(p2, r[6], fmaf(p1, r[3], p0 * r[0]));
      float o1 = fmaf(p2, r[7], fmaf(p1, r[4], p0 * r[1]));
      float o2 = fmaf(p2, r[8], fmaf(p1, r[5], p0 * r[2]));
      if (!translate_first) { o0 = o0 + tb[0]; o1 = o1 + tb[1]; o2 = o2 + tb[2]; }
      float* o = out + ((int64_t)b * n + i) * 3;
      o[0] = o0; o[1] = o1; o[2] = o2;
    }
  }
}
