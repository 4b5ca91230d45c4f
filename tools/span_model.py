"""Host model of the exclusive row spans of dm_window.hip (double precision), to size
what each kernel writes for a given workload geometry.  Not part of the product."""
import numpy as np

def corners(q0, q1, r0, r1, dmin, dmax, P):
  W, H, cx, cy, fx, fy, pitch, camh, yaw, tx, tz, res, wo, ho, mh, flip = P
  out = []
  for ci in range(8):
    z = dmax if ci & 1 else dmin
    r = (r1 - 1) if ci & 2 else r0
    q = (q1 - 1) if ci & 4 else q0
    yr = (H - 1 - r) if flip else r
    ax, ay = (q - cx) / fx, (yr - cy) / fy
    X, Y, Z = ax * z, ay * z, z
    c, s = np.cos(pitch), np.sin(pitch)
    # rotate about x by pitch (row-vector convention of utils.rotate)
    Rx = np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    v = np.array([X, Y, Z]) @ Rx
    v[1] += camh
    c, s = np.cos(yaw), np.sin(yaw)
    Ry = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    v = v @ Ry + np.array([tx, 0, tz])
    xf = v[0] / res + wo
    zf = v[2] / res + ho
    if flip: zf = (mh - 1) - zf
    out.append((xf + 0.5, zf + 0.5))
  return np.array(out)

def spans(cor, mh, mw):
  edges = []
  for axis in range(3):
    for k in range(4):
      low = (1 << axis) - 1
      ia = ((k & ~low) << 1) | (k & low)
      edges.append((ia, ia | (1 << axis)))
  x0 = max(0, int(np.floor(cor[:, 0].min())) - 2) & ~3
  x1 = min(mw, (int(np.floor(cor[:, 0].max())) + 3 + 3) & ~3)
  z0 = max(0, int(np.floor(cor[:, 1].min())) - 2)
  z1 = min(mh, int(np.floor(cor[:, 1].max())) + 3)
  res = {}
  for z in range(z0, z1):
    blo, bhi = z - 1, z + 2
    lo, hi = np.inf, -np.inf
    for a, b in edges:
      xa, za = cor[a]; xb, zb = cor[b]
      dz = zb - za
      if abs(dz) < 1e-6:
        if not (blo <= za <= bhi): continue
        t0, t1 = 0., 1.
      else:
        ta, tb = (blo - za) / dz, (bhi - za) / dz
        t0, t1 = max(min(ta, tb), 0.), min(max(ta, tb), 1.)
        if t0 > t1: continue
      for t in (t0, t1):
        x = xa + t * (xb - xa); lo = min(lo, x); hi = max(hi, x)
    if lo <= hi:
      l = max(int(np.floor(lo)) - 2, x0) & ~3
      r = (min(int(np.floor(hi)) + 3, x1) + 3) & ~3
      if l < r: res[z] = (l, r)
  return (x0, z0, x1 - x0, z1 - z0), res

def main():
  W, H, mh, mw = 640, 480, 512, 512
  hfov = np.radians(70.)
  cx, cy = W / 2., H / 2.
  fx = cx / np.tan(hfov / 2.); fy = fx
  rng = np.random.default_rng(0)
  tot = dict(bbox=0, span=0, excl=0, ubox=0)
  nf = 16
  for f in range(nf):
    tx, tz = rng.uniform(-1, 1, 2); yaw = rng.uniform(-np.pi, np.pi)
    P = (W, H, cx, cy, fx, fy, np.radians(-20.), 0.88, yaw, tx, tz, 0.03, mw / 2., mh / 2., mh, True)
    parts = []
    for pc in range(4):
      cor = corners(pc * 160, pc * 160 + 160, 0, H, 0.15, 5.05, P)
      parts.append(spans(cor, mh, mw))
    ux0 = min(p[0][0] for p in parts); ux1 = max(p[0][0] + p[0][2] for p in parts)
    uz0 = min(p[0][1] for p in parts); uz1 = max(p[0][1] + p[0][3] for p in parts)
    tot['ubox'] += (ux1 - ux0) * (uz1 - uz0)
    for i, (win, sp) in enumerate(parts):
      tot['bbox'] += win[2] * win[3]
      for z, (l, r) in sp.items():
        tot['span'] += r - l
        e0, e1 = l, r
        for j, (_, sq) in enumerate(parts):
          if j == i or z not in sq: continue
          lq, rq = sq[z]
          if rq <= e0 or lq >= e1: continue
          if lq <= e0: e0 = min(rq, e1)
          else: e1 = lq
        if e0 < e1: tot['excl'] += e1 - e0
  for k, v in tot.items(): print(k, v / nf, "cells per frame")

main()


def sheared_area(cor, mh, mw):
  """Area (cells) of the tightest parallelogram window around the 8 corners, sheared along
  z (rows shift in x) or along x (columns shift in z), 2 cells of slack, x aligned to 4."""
  x, z = cor[:, 0], cor[:, 1]
  best = None
  for major in ("z", "x"):
    a, b = (x, z) if major == "z" else (z, x)      # a = sheared coordinate, b = running coordinate
    # candidate shears: slopes of the hull's long edges -> try all pairs
    cands = [0.0]
    for i in range(8):
      for j in range(i + 1, 8):
        if abs(b[j] - b[i]) > 1.0:
          cands.append((a[j] - a[i]) / (b[j] - b[i]))
    b0 = np.floor(b.min()) - 2
    hb = np.floor(b.max()) + 3 - b0
    for sh in cands:
      d = a - sh * (b - b0)
      w = np.floor(d.max()) + 3 - (np.floor(d.min()) - 2) + (3 if major == "z" else 0) + abs(sh)
      area = w * hb
      if best is None or area < best[0]:
        best = (area, major, sh)
  return best


def shear_main():
  W, H, mh, mw = 640, 480, 512, 512
  hfov = np.radians(70.)
  cx, cy = W / 2., H / 2.
  fx = cx / np.tan(hfov / 2.); fy = fx
  rng = np.random.default_rng(0)
  tot_b = tot_s = 0; majors = {"z": 0, "x": 0}
  for f in range(16):
    tx, tz = rng.uniform(-1, 1, 2); yaw = rng.uniform(-np.pi, np.pi)
    P = (W, H, cx, cy, fx, fy, np.radians(-20.), 0.88, yaw, tx, tz, 0.03, mw / 2., mh / 2., mh, True)
    for pc in range(4):
      cor = corners(pc * 160, pc * 160 + 160, 0, H, 0.15, 5.05, P)
      bb = (np.floor(cor[:, 0].max()) + 3 - np.floor(cor[:, 0].min()) + 2 + 3) * (np.floor(cor[:, 1].max()) + 3 - np.floor(cor[:, 1].min()) + 2)
      a, major, sh = sheared_area(cor, mh, mw)
      tot_b += bb; tot_s += a; majors[major] += 1
  print("unclipped bbox cells/part %.0f   sheared window cells/part %.0f   (%s)" % (tot_b / 64, tot_s / 64, majors))

shear_main()


def bearing_main():
  """Window (bounding box) sizes if the pixels are partitioned by top-down bearing instead of
  by image column: each part's footprint is then an exact wedge."""
  W, H, mh, mw = 640, 480, 512, 512
  hfov = np.radians(70.); pitch = np.radians(-20.)
  cx, cy = W / 2., H / 2.
  fx = cx / np.tan(hfov / 2.); fy = fx
  c, s = np.cos(pitch), np.sin(pitch)
  # local z of a unit-depth pixel: z1 = p8 + p5*ay with (row-vector convention) p5 = -s ... use the model's matrices
  ays = np.array([((H - 1 - r) - cy) / fy for r in (0, H - 1)])
  Rx = np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
  den = np.array([(np.array([0, a, 1.0]) @ Rx)[2] for a in ays])
  dmin, dmax, res = 0.15, 5.05, 0.03
  axe = (np.array([0, W - 1]) - cx) / fx
  kap_out = [axe[0] / den.min(), axe[1] / den.min()]
  kaps = [kap_out[0], axe[0] / 2 / 1.0 * 0 + (160 - cx) / fx / den.mean(), 0.0, (480 - cx) / fx / den.mean(), kap_out[1]]
  rng = np.random.default_rng(0)
  tot = 0; n = 0; skew = 0
  for f in range(16):
    tx, tz = rng.uniform(-1, 1, 2); yaw = rng.uniform(-np.pi, np.pi)
    cyw, syw = np.cos(yaw), np.sin(yaw)
    for p in range(4):
      pts = []
      for k in (kaps[p], kaps[p + 1]):
        for z1 in (dmin * den.min(), dmax * den.max()):
          x1 = k * z1
          v = np.array([x1, 0, z1]) @ np.array([[cyw, 0, syw], [0, 1, 0], [-syw, 0, cyw]]) + np.array([tx, 0, tz])
          pts.append((v[0] / res + mw / 2 + 0.5, (mh - 1) - (v[2] / res + mh / 2) + 0.5))
      pts = np.array(pts)
      w = (np.floor(pts[:, 0].max()) + 3) - (np.floor(pts[:, 0].min()) - 2) + 3
      h = (np.floor(pts[:, 1].max()) + 3) - (np.floor(pts[:, 1].min()) - 2)
      tot += w * h; n += 1
    for k in kaps[1:4]:
      skew = max(skew, abs(fx * k * (den.max() - den.min())))
  print("bearing-partition bbox cells/part %.0f (column strips: 21516); halo columns needed: %.0f" % (tot / n, skew))

bearing_main()
