"""Parity campaign (GPU box, builder-run: too long for the test suite): the seeded parameter sweep of
tests/test_hip_parity.py::test_random_configurations_vs_oracle for many more seeds; the GPU result
against the CPU oracle, cells and heights bit for bit; prints every mismatch.
Usage: [MODE=1 ...] python tests/campaigns/parity_campaign.py [first seed] [count]

Modes (environment variables, combinable where it makes sense; results of every run of the round
in profiles/r03_parity_campaigns.log):
  DM_CAMPAIGN_ONE_PITCH  one pitch per batch + 0..8 forced column strips: the strip path
  DM_CAMPAIGN_CALLS      a third of the calls through orth_project_and_fuse (per-frame maps AND the
                         batch-fused map compared), a third through MapProjector.prepare
  DM_CAMPAIGN_FUSED      orth_project_fused: every frame into ONE map, strip width x frames per
                         workgroup forced at random, trajectories and unrelated poses
  DM_CAMPAIGN_BIG        240x320 .. 480x640 frames, maps up to 768x768, up to 70 frames
  DM_CAMPAIGN_FINE       fine resolutions on large maps: depth bands (forced)
  DM_CAMPAIGN_ODD        map widths that are not multiples of 4 (padded maps + copy-out)
  DM_CAMPAIGN_SUM        =1: reduction 'sum', =mean: reduction 'mean', =prod: 'prod' (one-hot classes: exact;
                         heights: order-dependent float sums, rtol = atol = 1e-5)
  DM_CAMPAIGN_OFFSETS    one map offset and one camera height per frame
  DM_CAMPAIGN_DC         depth maps of two or three channels (one map channel each), valid maps shared or per channel
  DM_CAMPAIGN_FILL_SPLIT where the strip path's fill duty of the rows outside a frame's union window runs, at random per
                         configuration (dm_debug_fill_split: under the loop / kernel head / combine kernel)
  DM_CAMPAIGN_FLOW       height maps through orth_project_and_flow (dm_orth_project_flow_f32), the one-kernel form switched
                         on at random: maps against the oracle, the flow grid against camera_affine_grid bit for bit
  DM_CAMPAIGN_SEMANTIC=0 no value maps;  DM_CAMPAIGN_EDGE=0 no NaN / inf depths, no missing truncations
  DM_CAMPAIGN_VERBOSE    print the configuration and the first differing cells"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dungeon_maps_amd as dmap
from oracle import oracle
from conftest import project_kwargs

def one(seed):
  rng = np.random.default_rng(50_000 + seed)
  B = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 40, 64, 70]))
  H, W = [(48, 64), (60, 80), (96, 128), (50, 70), (120, 160), (240, 320)][int(rng.integers(6))]
  if B * H * W > 3_000_000:
    B = max(1, 3_000_000 // (H * W))
  if SUM and B * 3 * H * W > 6_000_000:      # always one-hot counts (exact), never height sums
    B = max(1, 6_000_000 // (3 * H * W))
  if BIG:          # full-size frames: tall union windows, several passes of the row tables, 64-frame launches
    B = int(rng.choice([1, 2, 3, 8, 16, 33, 64, 70]))
    H, W = [(240, 320), (480, 640), (360, 480)][int(rng.integers(3))]
    if B * H * W > 20_000_000:
      B = max(1, 20_000_000 // (H * W))
  sizes = [(64, 64), (96, 128), (128, 96), (160, 160), (256, 256), (300, 200)]
  if BIG:
    sizes = [(256, 256), (512, 512), (384, 640), (640, 384), (512, 256), (768, 768)]
  if ODD:          # map widths that are not multiples of 4: padded maps + copy-out (dm_api.hip)
    sizes = [(65, 63), (97, 131), (128, 241), (255, 255), (301, 199), (100, 102)]
  mh, mw = sizes[int(rng.integers(6))]
  if FINE:      # long thin wedges: the windowed path needs depth bands (forced on small images)
    mh, mw = [(512, 512), (768, 1024), (1024, 1024), (1024, 640)][int(rng.integers(4))]
    if B * mh * mw > 12_000_000:
      B = max(1, 12_000_000 // (mh * mw))
  if rng.integers(2):
    rows = np.arange(H, dtype=np.float64).reshape(1, 1, H, 1)
    wall = rng.uniform(1.0, 6.0, size=(B, 1, 1, W // 8 + 1)).repeat(8, axis=3)[..., :W]
    floor = 0.9 / np.maximum(0.05, (rows - H * 0.45) / (H * 0.9))
    depth = np.broadcast_to(np.minimum(floor, wall).astype(np.float32), (B, 1, H, W)).copy()
  else:
    depth = rng.uniform(0.1, 8.0, size=(B, 1, H, W)).astype(np.float32)
  if EDGE and rng.integers(3) == 0:      # special values: never valid, never crash
    idx = rng.integers(0, depth.size, 12)
    depth.reshape(-1)[idx] = np.resize(np.array([np.nan, np.inf, -np.inf, 0.0, -1.0, 1e30], np.float32), 12)
  pose = np.stack([rng.uniform(-2, 2, B), rng.uniform(-2, 2, B), rng.uniform(-np.pi, np.pi, B)],
                  axis=1).astype(np.float32)
  per_frame = bool(rng.integers(2)) and not ONE_PITCH and not FUSED
  pitch = rng.uniform(-0.9, 0.5, size=B if per_frame else 1).astype(np.float32)
  camh = rng.uniform(0.2, 2.0, size=B if per_frame else 1).astype(np.float32)
  is_max = bool(rng.integers(4))
  res = float(rng.choice([0.02, 0.03, 0.05, 0.08, 0.1, 1.0 / 3, 0.0625]))
  if FINE:
    res = float(rng.choice([0.004, 0.006, 0.008, 0.01, 0.0125, 0.015]))
  cfg = dict(width=W, height=H, hfov=float(rng.uniform(0.6, 2.0)),
             vfov=None if rng.integers(2) else float(rng.uniform(0.5, 1.6)),
             cam_pitch=pitch, cam_height=camh,
             width_offset=float(mw / 2 + rng.uniform(-40, 40) * (8 if FINE else 1)),
             height_offset=float(mh / 2 + rng.uniform(-40, 40) * (8 if FINE else 1)),
             map_res=res, map_width=mw, map_height=mh,
             trunc_depth_min=None if EDGE and rng.integers(6) == 0 else float(rng.choice([0.0, 0.15, 0.5])),
             trunc_depth_max=None if EDGE and rng.integers(6) == 0 else float(rng.choice([1.5, 2.5, 5.05, 7.0])),
             trunc_height_max=None if rng.integers(3) else float(rng.uniform(0.2, 1.2)),
             clip_border=int(rng.choice([0, 0, 3, 9])),
             to_global=bool(rng.integers(2)), flip_h=bool(rng.integers(4)),
             fill_value=(-np.inf if is_max else np.inf) if rng.integers(3) else float(rng.uniform(-1, 1)),
             reduction="max" if is_max else "min")
  if OFFSETS:    # one map offset per frame (what MapBuilder's centre modes produce), camera heights per frame
    cfg["width_offset"] = (mw / 2 + rng.uniform(-40, 40, size=B)).astype(np.float32)
    cfg["height_offset"] = (mh / 2 + rng.uniform(-40, 40, size=B)).astype(np.float32)
    cfg["cam_height"] = rng.uniform(0.2, 2.0, size=B).astype(np.float32)
  valid = (rng.uniform(size=(B, 1, H, W)) > 0.1) if rng.integers(3) == 0 else None
  if DC:         # two or three depth channels, each projected into its own map channel; valid map per channel or shared
    dcs = int(rng.integers(2, 4))
    depth = np.concatenate([depth] + [rng.uniform(0.1, 8.0, size=(B, 1, H, W)).astype(np.float32)
                                      for _ in range(dcs - 1)], axis=1)
    if valid is not None and rng.integers(2):
      valid = rng.uniform(size=(B, dcs, H, W)) > 0.1
  value = None
  if SUM:       # point counts per class: small integers, exact in float32 whatever the order
    cfg["reduction"] = SUM_KIND if SUM_KIND in ("mean", "prod") else "sum"
    cfg["fill_value"] = float(rng.choice([0.0, 1.0, 5.0]))
  C = int(rng.choice([0, 0, 3, 9])) if SEMANTIC and not FUSED and not DC else 0
  if DC and SEMANTIC and not FUSED and not SUM and rng.integers(2):
    C = depth.shape[1]           # a value channel per depth channel (dc == vc: every channel its own cells)
  if SUM:
    C = 3
  if C and B * C * H * W <= 6_000_000:      # value maps: one-hot labels or random reals
    if SUM or rng.integers(2):
      value = np.eye(C, dtype=np.float32)[rng.integers(0, C, size=(B, H, W))].transpose(0, 3, 1, 2).copy()
    else:
      value = rng.normal(size=(B, C, H, W)).astype(np.float32)
  proj = dmap.MapProjector(**cfg)
  get_h = value is not None
  if FUSED:      # dm_orth_project_fused_f32: every frame into ONE map, strips x frame groups forced at random
    if rng.integers(2):      # a trajectory (frame groups then share a window), else unrelated poses
      k = np.arange(B, dtype=np.float32)
      pose = np.stack([pose[0, 0] + 0.02 * k, pose[0, 1] + 0.01 * k, pose[0, 2] + 0.01 * k], axis=1).astype(np.float32)
    LIB.dm_debug_force_fused_split(int(rng.choice([0, 16, 20, 40, 80, 160])), int(rng.choice([0, 1, 2, 4, 8])))
    try:
      fo, fm = proj.orth_project_fused(torch.from_numpy(depth).cuda(),
                                       valid_map=None if valid is None else torch.from_numpy(valid).cuda(),
                                       cam_pose=pose)
    finally:
      LIB.dm_debug_force_fused_split(0, 0)
    split = (ctypes.c_int32 * 4)()
    LIB.dm_debug_last_fused_split(split)
    STATS["strip"] += split[1] > 0
    kw = dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose)
    want = oracle.orth_project(depth, value_map=None, valid_map=valid, get_height_map=False, nthreads=16, **kw)
    wf = want[0].max(axis=0) if cfg["reduction"] == "max" else want[0].min(axis=0)
    wm = want[1].any(axis=0)
    gf, gm = fo.cpu().numpy(), fm.cpu().numpy()
    bad_m = int((gm != wm).sum())
    bad_v = int((~((gf == wf) | (np.isnan(gf) & np.isnan(wf)))).sum())
    return bad_m, bad_v, (B, H, W, mh, mw, res, tuple(split))
  if ONE_PITCH:      # any number of column strips, whatever the cost model says (the strip path's own hook)
    LIB.dm_debug_force_strips(int(rng.integers(0, 9)))
  if FILL_SPLIT:
    LIB.dm_debug_fill_split(int(rng.choice([-1, 0, 3, 5, 8])))
  # (the strip path's shared groups through compact planes -- the default -- or through slabs + lists, at random)
  LIB.dm_debug_planes(int(rng.choice([-1, -1, 0])))
  d_dev = torch.from_numpy(depth).cuda()
  v_dev = None if value is None else torch.from_numpy(value).cuda()
  m_dev = None if valid is None else torch.from_numpy(valid).cuda()
  how = int(rng.integers(3)) if CALLS else 0
  fused_pair = None
  flow_bad = 0
  if how == 1 and cfg["reduction"] in ("max", "min"):      # per-frame maps + the batch-fused map in one call
    top, msk, fo, fm = proj.orth_project_and_fuse(d_dev, value_map=v_dev, valid_map=m_dev, cam_pose=pose)
    outs = (top, msk)
    fused_pair = (fo.cpu().numpy(), fm.cpu().numpy())
    get_h = False
  elif how == 2:      # prepared frames (poses in a device buffer), where the strip path takes the call
    try:
      prep = proj.prepare(B, cam_pose=pose, value_channels=0 if value is None else value.shape[1],
                          valid_channels=0 if valid is None else valid.shape[1],
                          depth_channels=depth.shape[1])
    except Exception:
      prep = None
    if prep is not None:
      STATS["prepared"] = STATS.get("prepared", 0) + 1
      outs = prep.orth_project(d_dev, value_map=v_dev, valid_map=m_dev, get_height_map=get_h)
    else:
      outs = proj.orth_project(d_dev, value_map=v_dev, valid_map=m_dev, cam_pose=pose, get_height_map=get_h)
  elif FLOW and value is None and cfg["reduction"] in ("max", "min"):
    tp = np.stack([rng.uniform(-0.3, 0.3, B), rng.uniform(-0.3, 0.3, B), rng.uniform(-0.5, 0.5, B)], axis=1).astype(np.float32)
    LIB.dm_debug_flow_fused(int(rng.integers(2)))
    try:
      top, msk, grid = proj.orth_project_and_flow(d_dev, tp, valid_map=m_dev, cam_pose=pose)
      STATS["flow_fused"] = STATS.get("flow_fused", 0) + LIB.dm_debug_last_flow_fused()
    finally:
      LIB.dm_debug_flow_fused(0)
    alone = proj.camera_affine_grid(d_dev, tp)
    if not torch.equal(grid.view(torch.int32), alone.view(torch.int32)):
      flow_bad = int((grid.view(torch.int32) != alone.view(torch.int32)).sum())
    outs = (top, msk)
    get_h = False
  else:
    outs = proj.orth_project(d_dev, value_map=v_dev, valid_map=m_dev, cam_pose=pose, get_height_map=get_h)
  split = (ctypes.c_int32 * 4)()
  LIB.dm_debug_last_split(split)
  STATS["banded"] += split[2] > 1
  STATS["generic"] += split[0] == 0
  STATS["strip"] += LIB.dm_debug_last_path() == 2
  kw = dict(project_kwargs(cfg, oracle.camera_intrinsics), cam_pose=pose)
  want = oracle.orth_project(depth, value_map=value, valid_map=valid, get_height_map=get_h,
                             nthreads=16, **kw)
  got = [o.cpu().numpy() for o in outs]
  bad_m = int((got[1] != want[1]).sum())
  bad_v = flow_bad
  if fused_pair is not None:
    wf = want[0].max(axis=0) if cfg["reduction"] == "max" else want[0].min(axis=0)
    bad_m += int((fused_pair[1] != want[1].any(axis=0)).sum())
    bad_v += int((~((fused_pair[0] == wf) | (np.isnan(fused_pair[0]) & np.isnan(wf)))).sum())
  if os.environ.get("DM_CAMPAIGN_VERBOSE"):
    d = ~((got[0] == want[0]) | (np.isnan(got[0]) & np.isnan(want[0])))
    idx = np.argwhere(d)[:12]
    print("cfg", {k: v for k, v in cfg.items() if k not in ("cam_pitch", "cam_height")})
    for i in idx:
      print("  cell", tuple(i), "got", got[0][tuple(i)], "want", want[0][tuple(i)])
    print("  differing cells", int(d.sum()), "per frame", d.sum(axis=(1, 2, 3))[:16])
  exact = not (SUM and value is None)     # sums of heights are order dependent in float32
  for a, b in ((got[0], want[0]),) + (((got[2], np.ascontiguousarray(want[2])),) if get_h else ()):
    if exact:
      bad_v += int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum())
    else:
      bad_v += int((~(np.isclose(a, b, rtol=1e-5, atol=1e-5) | (np.isnan(a) & np.isnan(b)) |
                      (a == b))).sum())
  if not exact and bad_m <= 1e-4 * got[1].size:
    bad_m = 0                               # a sum that lands on the fill value by rounding
  return bad_m, bad_v, (B, H, W, mh, mw, res, C)

_MODES = ("ODD", "SUM", "FINE", "ONE_PITCH", "OFFSETS", "DC", "BIG", "CALLS", "FUSED", "FILL_SPLIT", "FLOW")
STATS = {"banded": 0, "generic": 0, "strip": 0}
from dungeon_maps_amd import _native
LIB = _native.lib()


def configure(env=None):
  """Set the campaign's modes from DM_CAMPAIGN_* variables (`env`: a dict, default os.environ); the GPU test
  suite runs a slice of every mode through this (tests/test_hip_parity.py)."""
  env = os.environ if env is None else env
  g = globals()
  for name in _MODES:
    g[name] = env.get("DM_CAMPAIGN_" + name, "0") != "0"
  g["SUM_KIND"] = env.get("DM_CAMPAIGN_SUM", "0")
  g["SEMANTIC"] = env.get("DM_CAMPAIGN_SEMANTIC", "1") != "0"
  g["EDGE"] = env.get("DM_CAMPAIGN_EDGE", "1") != "0"      # NaN/inf depths, missing truncations
  for k in list(STATS):
    STATS[k] = 0
  LIB.dm_debug_force_bands(1 if g["FINE"] else 0)


def reset_switches():
  """Back to the library's defaults (the modes leave per-thread debug switches set)."""
  LIB.dm_debug_force_bands(0); LIB.dm_debug_force_strips(0); LIB.dm_debug_fill_split(-1); LIB.dm_debug_flow_fused(0)
  LIB.dm_debug_planes(-1)
  LIB.dm_debug_force_fused_split(0, 0)


configure()


def main():
  first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
  count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
  bad = 0
  for s in range(first, first + count):
    try:
      bm, bv, shape = one(s)
    except Exception:
      print("EXCEPTION at seed", s, flush=True)
      raise
    if bm or bv:
      bad += 1
      print("MISMATCH seed", s, shape, "mask cells", bm, "map cells", bv, flush=True)
    if (s - first) % 50 == 49:
      print("  ... %d configurations, %d with mismatches" % (s - first + 1, bad), flush=True)
  print("done: %d configurations, %d with mismatches (%d took the strip path, %d depth bands, %d the generic path%s)"
        % (count, bad, STATS["strip"], STATS["banded"], STATS["generic"],
           (", %d through prepared frames" % STATS["prepared"] if "prepared" in STATS else "") +
           (", %d flow grids from the projection kernel" % STATS["flow_fused"] if "flow_fused" in STATS else "")))


if __name__ == "__main__":
  main()
