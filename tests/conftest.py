import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
  """Return (arrays, cfg): cfg_* scalars become a dict, NaN -> None."""
  z = np.load(os.path.join(GOLDEN, name + ".npz"))
  arrays, cfg = {}, {}
  for k in z.files:
    if k.startswith("cfg_"):
      v = z[k]
      if v.ndim == 0:
        v = v.item()
        if isinstance(v, float) and np.isnan(v):
          v = None
      else:
        v = v.tolist()
      cfg[k[4:]] = v
    else:
      arrays[k] = z[k]
  return arrays, cfg


def load_sweep(name="g11_parameter_sweep_56x40"):
  """The parameter-sweep fixture as a list of (arrays, cfg) pairs, one per configuration."""
  z = np.load(os.path.join(GOLDEN, name + ".npz"))
  out = {}
  for key in z.files:
    k, rest = key.split("_", 1)
    arrays, cfg = out.setdefault(int(k[1:]), ({}, {}))
    if rest.startswith("cfg_"):
      v = z[key]
      v = v.item() if v.ndim == 0 else v.tolist()
      if isinstance(v, float) and np.isnan(v):
        v = None
      cfg[rest[4:]] = v
    else:
      arrays[rest] = z[key]
  return [out[k] for k in sorted(out)]


def project_kwargs(cfg, intrinsics_fn):
  """Split a MapProjector config into orth_project keyword arguments the way
  MapProjector.orth_project forwards them (reference maps.py:1438-1465)."""
  cx, cy, fx, fy = intrinsics_fn(cfg["width"], cfg["height"], cfg["hfov"], cfg.get("vfov"))
  kw = dict(center_x=cx, center_y=cy, focal_x=fx, focal_y=fy)
  for k in ("cam_pose", "width_offset", "height_offset", "cam_pitch", "cam_height",
            "map_res", "map_width", "map_height", "trunc_depth_min", "trunc_depth_max",
            "trunc_height_max", "clip_border", "reduction"):
    kw[k] = cfg.get(k)
  kw["to_global"] = bool(cfg.get("to_global", False))
  kw["flip_h"] = bool(cfg.get("flip_h", True))
  kw["fill_value"] = cfg.get("fill_value", -np.inf) if "fill_value" in cfg else -np.inf
  return kw


@pytest.fixture(scope="session")
def oracle():
  from oracle import oracle as orc
  orc.build()
  return orc


def assert_masks_equal_away_from_fill(got_mask, want_mask, want_values, fill, exact=False, what=""):
  """Masks are integer work ("the cell changed", reference utils.py:489-491): equal, except that an
  order-dependent float32 sum / mean / product may land on either side of the fill value where
  the reference's own result sits within rounding of it (`exact`: not even there -- sums that are
  exact in float32, values that cannot cancel)."""
  differ = np.asarray(got_mask) != np.asarray(want_mask)
  if exact:
    assert not differ.any(), f"{what}: {int(differ.sum())} mask cells differ"
    return
  fill = 0.0 if fill is None else float(fill)
  near = np.abs(np.asarray(want_values, dtype=np.float64) - fill) <= 1e-5 * max(1.0, abs(fill))
  bad = differ & ~near
  assert not bad.any(), f"{what}: {int(bad.sum())} mask cells differ away from the fill value"
