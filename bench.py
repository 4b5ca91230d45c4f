#!/usr/bin/env python3
"""Headline benchmark: depth frames/s projected + fused on MI355X.

One step = one pass of the hot path over one batch of synthetic frames per GPU:

  1. orth_project  B=64 x (640x480) depth -> (64, 1, 512, 512) height maps+masks, with a NEW
     set of camera poses on every step, passed the way the reference's callers pass them
     (MapProjector.orth_project(depth, cam_pose=...), maps.py:1406-1465)
     (BASELINE.json configs[1]; the kernel sequence the roofline is quoted on)
  2. fuse          per-rank partial global map = max over the rank's frames
  3. all-reduce    element-wise max of the partial maps over RCCL (N > 1 only)
  4. mask          of the fused map

N ranks each process their own 64 frames (weak scaling); `value` is the
whole-job rate = N*B*K / max-over-ranks elapsed.  Inputs are synthetic,
seeded, and resident in HBM before the timed region.

The working set is larger than the chip's 256 MiB Infinity Cache on purpose: the steps rotate
over ROT depth batches and ROT caller-owned output sets (4 x 79 MB in, 4 x 84 MB out at cfg2),
so a line is touched again only after > 256 MiB of other traffic and every byte of a step is
served by HBM -- in the timed loop and in the roofline's launch measurement alike.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: "roofline" (HBM, algorithmic bytes of step 1 / its HIP-event time on
the launch stream) and "cpu_baseline" (the oracle timed on this box's host
cores on a bounded sample; N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MIX_CEILING_GBS = 5330.0  # measured: the best plain kernel moving cfg2's read / write mix (profiles/r05_traffic_model.log)

WORKLOADS = {
    # name: (B, H, W, mh, mw, value channels)
    "cfg1": (1, 240, 320, 256, 256, 0),
    "cfg2": (64, 480, 640, 512, 512, 0),
    "cfg3": (64, 480, 640, 512, 512, 40),
    # per rank 64 frames of one trajectory fused straight into ONE 1024x1024 global map
    # (no per-frame maps), then the cross-rank max all-reduce: BASELINE configs[3]
    "cfg4": (64, 480, 640, 1024, 1024, 0),
    # per rank 16 frames of 1280x960 -> 2048x2048, the ego-motion flow grid of the same depth
    # maps and a crop of every map around its camera's cell: BASELINE configs[4]
    "cfg5": (16, 960, 1280, 2048, 2048, 0),
}
CFG5_CROP = 1024          # crop side (cells) of the cfg5 workload's TopdownMap.select leg
ROT_TARGET_BYTES = 600 << 20      # rotate until a step's buffers come around after > 2 x 256 MiB


def algorithmic_bytes(B, H, W, mh, mw, C, fused_only=False):
  """read every input once, write every output once."""
  depth = B * H * W * 4
  if fused_only:
    return depth + mh * mw * (4 + 1)
  if C == 0:
    return depth + B * mh * mw * (4 + 1)
  return depth + B * C * H * W * 4 + B * C * mh * mw * (4 + 1)     # object map, no height map


def synthetic_inputs(B, H, W, C, seed, device, scene):
  g = torch.Generator().manual_seed(seed)
  if scene:
    depth = scene_depth(B, H, W, g)
  else:
    depth = torch.empty(B, 1, H, W).uniform_(0.1, 10.0, generator=g)
  pose = torch.empty(B, 3).uniform_(-1, 1, generator=g)
  pose[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=g)
  value = None
  if C:
    labels = torch.randint(0, C, (B, H, W), generator=g)
    value = torch.nn.functional.one_hot(labels, C).permute(0, 3, 1, 2).float().contiguous()
  return depth, pose, value


def scene_depth(B, H, W, g, hfov=np.radians(70.), pitch=np.radians(-20.), cam_h=0.88):
  """Floor plane + box walls (SURVEY 8d 'scene-like' variant)."""
  cx, cy = W / 2., H / 2.
  fx = cx / np.tan(hfov / 2.)
  r = torch.arange(H, dtype=torch.float64).view(1, H, 1)
  yn = ((H - 1) - r - cy) / fx
  ky = np.cos(pitch) * yn + np.sin(pitch)
  floor = torch.where(ky < -1e-6, -cam_h / ky, torch.full_like(ky, 10.0)).expand(B, H, W)
  walls = torch.empty(B, 1, 8).uniform_(1.5, 6.0, generator=g).double()
  walls = walls.repeat_interleave(W // 8, dim=2).expand(B, H, W)
  d = torch.minimum(floor, walls).clamp(0.1, 10.0)
  return d.float().unsqueeze(1).contiguous()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=20)
  ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
  ap.add_argument("--depth", default="uniform", choices=["uniform", "scene"])
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-other-configs", action="store_true",
                  help="skip the other BASELINE.json configs (reported outside the timed region)")
  ap.add_argument("--no-two-streams", action="store_true",
                  help="skip the two-stream figure (profiling runs: its overlapping kernels would blur the "
                       "per-kernel averages)")
  ap.add_argument("--cpu-seconds", type=float, default=12.0)
  ap.add_argument("--event-every", type=int, default=16,
                  help="bracket every n-th timed step with HIP events (an event record costs "
                       "a few us of stream time, so not every step carries one)")
  ap.add_argument("--rotate", type=int, default=0,
                  help="depth batches / output sets the steps rotate over (0: as many as it takes for a "
                       "step's buffers to come around after more than twice the Infinity Cache; 1: one "
                       "set, everything cache resident -- rounds 1-3 measured that)")
  args = ap.parse_args()

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus:
    # one process per GPU: --gpus N runs under `python -m torch.distributed.run --nproc-per-node N`
    print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with "
          f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} "
          f"--master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`", file=sys.stderr)
    raise SystemExit(2)
  # (rehearsals on a one-GPU box: DM_BENCH_BACKEND=gloo lets several ranks share the device)
  backend = os.environ.get("DM_BENCH_BACKEND", "nccl")
  if backend != "nccl":
    local_rank %= max(1, torch.cuda.device_count())
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  dist = None
  if world > 1 or "RANK" in os.environ:     # launched by torch.distributed.run (any N)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
      dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
      dist.init_process_group(backend, rank=rank, world_size=world)

  import dungeon_maps_amd as dmap
  from dungeon_maps_amd import _native
  lib = _native.lib()   # the HIP library must be present: no fallback

  B, H, W, mh, mw, C = WORKLOADS[args.workload]
  fused_only = args.workload == "cfg4"
  cfg5 = args.workload == "cfg5"
  fill = 0.0 if C else -np.inf
  C_out = C if C else 1
  alg = algorithmic_bytes(B, H, W, mh, mw, C, fused_only)
  # How many input / output sets the steps rotate over: a set comes around again only after more
  # than ROT_TARGET_BYTES of other traffic, so nothing of it is left in the 256 MiB Infinity Cache.
  rot = args.rotate if args.rotate > 0 else max(1, min(8, -(-ROT_TARGET_BYTES // alg) + 1))
  depth_sets, value_sets = [], []
  depth = pose = value = None
  for j in range(rot):
    d_j, pose_j, v_j = synthetic_inputs(B, H, W, C, 1234 + rank + 1000 * j, dev, args.depth == "scene")
    if j == 0:
      depth, pose, value = d_j, pose_j, v_j          # (host copies: the CPU baseline's sample)
    depth_sets.append(d_j.to(dev))
    value_sets.append(None if v_j is None else v_j.to(dev))
  # the camera moves: every step gets another set of poses (host tensors, as a caller holds them)
  gp = torch.Generator().manual_seed(4321 + rank)
  pose_sets = [pose]
  for _ in range(7):
    q = torch.empty(B, 3).uniform_(-1, 1, generator=gp)
    q[:, 2] = torch.empty(B).uniform_(-np.pi, np.pi, generator=gp)
    pose_sets.append(q)
  proj = dmap.MapProjector(
      width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
      width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
      trunc_depth_min=0.15, trunc_depth_max=5.05, clip_border=0, to_global=True,
      fill_value=fill, reduction="max")
  # caller-owned output sets (`out=` / `fused_out=`): no allocation per call, and ROT of them so
  # that the maps a step writes are not the cache-resident maps of the step before
  out_sets = fused_sets = None
  if not fused_only:
    out_sets = [(torch.empty((B, C_out, mh, mw), dtype=torch.float32, device=dev),
                 torch.empty((B, C_out, mh, mw), dtype=torch.bool, device=dev)) for _ in range(rot)]
    fused_sets = [(torch.empty((C_out, mh, mw), dtype=torch.float32, device=dev),
                   torch.empty((C_out, mh, mw), dtype=torch.bool, device=dev)) for _ in range(rot)]
  if cfg5:
    flow_tp = torch.tensor([0.05, 0.1, 0.02])
    # crop centres = the cameras' cells (flip_h: the map's rows run against z), per pose set
    centers = []
    for q in pose_sets:
      cxs = q[:, 0] / 0.03 + mw / 2.
      czs = (mh - 1) - (q[:, 1] / 0.03 + mh / 2.)
      centers.append(torch.stack((cxs, czs), dim=1).to(dev))

  # HIP events on the launch stream (torch's current stream): before the call and,
  # through the library's measurement hook, right after the kernels that produce the
  # per-frame maps + masks (i.e. before the batch fuse).
  # the timed steps that carry their own pair of events (a cross-check of the back-to-back figure, not the
  # source of any reported rate): every event_every-th, starting in the middle of the first stretch -- an
  # event record is a stream operation of its own (~1.5-4 us), so few steps carry one
  bracketed_steps = set(range(min(args.event_every // 2, max(0, args.steps - 1)), args.steps, max(1, args.event_every)))
  ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
  ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
  for e in ev_a + ev_b:
    e.record()                 # materialise the hipEvent_t handles
  torch.cuda.synchronize()

  if fused_only:     # one trajectory, shared offsets
    k = torch.arange(B, dtype=torch.float32) + rank * B
    pose = torch.stack((0.02 * k, 0.01 * k, 0.01 * k), dim=1)
    pose_sets = [pose]
  calls = {"n": 0}      # steps issued so far (warm-up included): picks the step's pose / buffer sets

  # Cross-rank fuse: every step writes its partial global map into a slot of a ring;
  # once per RING steps ONE RCCL all-reduce(max) fuses the whole ring (fewer, larger
  # collectives: xGMI rings are latency bound at 1 MiB) on RCCL's own stream while the
  # next steps project; the masks of the reduced maps are recomputed afterwards.
  # Ring size: half of the timed steps, at most 32 -- the LAST collective of a run has nothing to hide under, so a
  # ring as long as the run (32 slots for the harness's 20 steps) would leave all of the run's maps to one exposed
  # all-reduce behind the last step; many short rings cost the host an enqueue + wait + mask launch each (one rank
  # over RCCL on one GPU, 20 steps: rings of 7 / 10 / 20 / 32 within the noise of each other, 59-68 us per step).
  RING_MAX = max(1, int(os.environ.get("DM_BENCH_RING", str(min(32, max(1, args.steps // 2))))))
  ring = ring_mask = ring_slots = None
  if dist is not None:
    ring = [torch.empty((RING_MAX, C_out, mh, mw), dtype=torch.float32, device=dev) for _ in range(2)]
    ring_mask = [torch.empty((RING_MAX, C_out, mh, mw), dtype=torch.bool, device=dev) for _ in range(2)]
    # (the slots as views made once: two indexing calls per step are 6 us of host time)
    ring_slots = [[(ring[b][k], ring_mask[b][k]) for k in range(RING_MAX)] for b in range(2)]
  state = {"slot": 0, "buf": 0, "pending": None, "ring": RING_MAX}

  def finish_reduce():
    if state["pending"] is not None:
      work, buf, n = state["pending"]
      state["pending"] = None
      work.wait()                                   # compute stream waits for the collective
      return dmap.mask_from_map(ring[buf][:n], fill)
    return None

  def flush_ring():
    n = state["slot"]
    if n == 0:
      return
    finish_reduce()
    buf = state["buf"]
    work = dist.all_reduce(ring[buf][:n], op=dist.ReduceOp.MAX, async_op=True)
    state["pending"] = (work, buf, n)
    state["buf"], state["slot"] = 1 - buf, 0

  def next_slot():
    buf, slot = state["buf"], state["slot"]
    state["slot"] = slot + 1
    return ring_slots[buf][slot]

  def step(i=None):
    if i is not None and i not in bracketed_steps:
      i = None                   # untimed by events (the wall clock still covers it)
    n = calls["n"]
    calls["n"] = n + 1
    pose = pose_sets[n % len(pose_sets)]
    depth_d, value_d = depth_sets[n % rot], value_sets[n % rot]
    if fused_only:
      if i is not None:
        ev_a[i].record()
      if dist is not None:
        fused, fmask = next_slot()
        fused.fill_(fill)          # running-map semantics: start from fill, accumulate
        fused, fmask = proj.orth_project_fused(depth_d, cam_pose=pose, out=fused)
      else:
        fused, fmask = proj.orth_project_fused(depth_d, cam_pose=pose)
      if i is not None:
        ev_b[i].record()
      if dist is not None and state["slot"] == state["ring"]:
        flush_ring()
      return fused, fmask, fused, fmask
    if i is not None:
      lib.dm_debug_record_before_projection(ev_a[i].cuda_event)
      lib.dm_debug_record_after_projection(ev_b[i].cuda_event)
    if cfg5:
      # per-frame maps + masks and the ego-motion flow grid of the same depth maps from ONE native
      # call, then the crop of every map around its camera's cell
      top, mask, grid = proj.orth_project_and_flow(depth_d, flow_tp, cam_pose=pose, out=out_sets[n % rot])
      crop, crop_mask = dmap.functional.crop_nearest(top, centers[n % len(pose_sets)], CFG5_CROP, CFG5_CROP,
                                                     fill_value=fill, mask=mask)
      del grid, crop, crop_mask
      return top, mask, top[0], mask[0]
    # per-frame maps + masks and this rank's partial global map, one launch sequence
    top, mask, fused, fmask = proj.orth_project_and_fuse(
        depth_d, value_map=value_d, cam_pose=pose, out=out_sets[n % rot],
        fused_out=next_slot() if dist is not None else fused_sets[n % rot])
    if dist is not None and state["slot"] == state["ring"]:
      flush_ring()                                      # RCCL, element-wise max
    return top, mask, fused, fmask

  def barrier():
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize()

  def timed_run(nsteps, nwarm, ring_steps, trace=None):
    """W warm-up steps, then exactly K timed steps between barrier + synchronize on both sides:
    (elapsed seconds of the slowest rank, host seconds to issue the steps, per-rank ms per step,
    the last step's outputs, device allocations inside the timed loop)."""
    state["ring"] = ring_steps
    # the warm-up runs the timed loop's exact body: the steps' outputs are bound to `out` and every
    # n-th step is bracketed by events
    out = None
    for j in range(nwarm):
      out = step(j % max(1, nsteps))
    if dist is not None:
      flush_ring()
      finish_reduce()
    barrier()
    segs0 = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0)
    t0 = time.perf_counter()
    for i in range(nsteps):
      out = step(i)
      if trace is not None:
        trace.append(time.perf_counter())
    enqueue_s = time.perf_counter() - t0      # host time to issue the steps (<= elapsed)
    segs1 = torch.cuda.memory_stats(dev).get("segment.all.allocated", 0)
    if dist is not None:            # every step's all-reduce + mask is inside the timed region
      flush_ring()
      finish_reduce()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / nsteps * 1e3]
    if dist is not None:
      t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
      every = [torch.zeros_like(t) for _ in range(world)]
      dist.all_gather(every, t)
      rank_ms = [float(x.item()) / nsteps * 1e3 for x in every]
      elapsed = max(float(x.item()) for x in every)         # the slowest rank's clock
    return elapsed, enqueue_s, rank_ms, out, segs1 - segs0, t0

  trace = [] if os.environ.get("DM_BENCH_TRACE") else None
  elapsed, enqueue_s, rank_ms, out, new_segments, t0 = timed_run(args.steps, args.warmup, RING_MAX, trace)

  # checksum of the last step's fused map (a one-rank RCCL run must reproduce the plain run), taken
  # before the measurements below write into the rotating output sets again
  fz, fm = out[2], out[3]
  finite = torch.isfinite(fz) & fm
  fused_checksum = {
      "cells": int(fm.sum().item()),
      "sum": float(torch.where(finite, fz, torch.zeros_like(fz)).double().sum().item()),
  }
  del fz, fm, finite
  proj_ms = np.array([ev_a[i].elapsed_time(ev_b[i]) for i in sorted(bracketed_steps)])
  bracketed_s = float(np.mean(proj_ms)) * 1e-3
  last_pose = pose_sets[(calls["n"] - 1) % len(pose_sets)]      # the poses of `out`

  def back_to_back(fn, n_b2b=64):
    """fn(j) n_b2b times between ONE pair of HIP events on the launch stream, seconds per call.  (A
    HIP event between two kernels is a stream operation of its own, ~1.5 us each way: bracketing
    every launch sequence would time the events too.  This is the figure rocprofv3's per-kernel
    averages add up to, profiles/.)"""
    for j in range(16):      # (enough calls for the clocks to settle: the first dozen after an idle gap run ~2 us slower)
      fn(j)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for j in range(n_b2b):
      fn(16 + j)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n_b2b

  step_s = None
  if fused_only:
    kernel_s = back_to_back(lambda j: proj.orth_project_fused(depth_sets[j % rot], cam_pose=pose_sets[j % len(pose_sets)]))
    b2b_note = ("64 orth_project_fused calls back to back between one pair of HIP events on the launch "
                "stream, total / 64")
  else:
    # The launch sequence of the per-frame maps (k_strip_scatter + k_strip_combine) exactly as the
    # timed steps launch it: the kernel variant a call with a batch fuse takes (the streaming one:
    # non-temporal fill stores and depth loads; forced through the library's measurement switch -- a
    # plain call of this size picks it by itself, a smaller one would not), the rotating depth batches, and
    # output blocks that rotate too -- plain orth_project allocates its outputs, so the last `rot`
    # results are kept alive and the allocator has to hand out another block every call.
    keep = [None] * (rot + 1)

    def plain(j):
      keep[j % len(keep)] = None
      keep[j % len(keep)] = proj.orth_project(depth_sets[j % rot], value_map=value_sets[j % rot],
                                              cam_pose=pose_sets[j % len(pose_sets)])
    lib.dm_debug_force_nt_fill(1)
    try:
      kernel_s = back_to_back(plain, 64 if alg < (1 << 30) else 8)
    finally:
      lib.dm_debug_force_nt_fill(-1)
    height_s = None
    if C:      # BASELINE.md section 4 counts cfg3 WITH the height map (maps.py:332-350, what MapBuilder.plot asks for)
      def plain_h(j):
        keep[j % len(keep)] = None
        keep[j % len(keep)] = proj.orth_project(depth_sets[j % rot], value_map=value_sets[j % rot],
                                                cam_pose=pose_sets[j % len(pose_sets)], get_height_map=True)
      lib.dm_debug_force_nt_fill(1)
      try:
        height_s = back_to_back(plain_h, 8)
      finally:
        lib.dm_debug_force_nt_fill(-1)
    del keep
    b2b_note = ("64 orth_project calls (a new set of poses, another depth batch and other output blocks "
                "each; the kernel variant of the timed steps: non-temporal fill stores) back to back "
                "between one pair of HIP events on the launch stream, total / 64")
    # the whole step's device time the same way: the timed loop's own call (project + batch fuse)
    step_s = None if cfg5 else back_to_back(lambda j: proj.orth_project_and_fuse(
        depth_sets[j % rot], value_map=value_sets[j % rot], cam_pose=pose_sets[j % len(pose_sets)],
        out=out_sets[j % rot], fused_out=fused_sets[j % rot]), 64 if alg < (1 << 30) else 8)
  achieved = alg / kernel_s / 1e9

  # HBM bytes of one launch sequence from the PMC counters: bench.py cannot run rocprofv3 on
  # itself, so it quotes the committed PMC summary of this very command (profiles/) -- but only
  # when that summary was taken on the library build that is loaded now (md5 of the .so)
  traffic, traffic_note = committed_traffic(args.workload, _native.LIB_PATH)

  result = {
      "metric": "depth frames/sec projected+fused, B=64 640x480->512x512",
      "value": world * B * args.steps / elapsed,
      "unit": "frames/s",
      "n_gpus": world,
      "steps": args.steps,
      "warmup": args.warmup,
      "ms_per_step": elapsed / args.steps * 1e3,
      "host_ms_per_step": enqueue_s / args.steps * 1e3,
      "higher_is_better": True,
      "scaling": "weak",
      "vs_baseline": None,
      "dtype": "f32",
      "data": "synthetic",
      "ranks": {"world_size": world,
                "backend": (dist.get_backend() if dist is not None else None),
                "launched_by_torch_distributed": dist is not None,
                "ms_per_step_min": min(rank_ms), "ms_per_step_max": max(rank_ms)},
      "config": {
          "workload": f"{args.workload}: B={B}/GPU, {W}x{H} depth -> {mw}x{mh} "
                      f"{'height map' if not C else f'{C}-class object map'}, "
                      f"depth={args.depth}, project + fuse(max over frames)"
                      f"{' + RCCL all-reduce(max)' if world > 1 else ''} + mask",
          "frames_per_gpu": B, "global_frames": world * B,
          "working_set": f"{rot} depth batches and {rot} caller-owned output sets in rotation "
                         f"({rot * alg / 2**20:.0f} MiB live: beyond the 256 MiB Infinity Cache)"
                         if rot > 1 else "one depth batch, one output set",
          "parallelism": f"frames sharded {world}-way; fused maps all-reduced ({'RCCL' if backend == 'nccl' else backend} max) "
                         f"{RING_MAX} steps at a time, overlapped with the next steps"
                         if dist is not None else "single GPU",
      },
      "roofline": {
          "bound": "hbm",
          "achieved": achieved,
          "peak": HBM_PEAK_GBS,
          "unit": "GB/s",
          "frac": achieved / HBM_PEAK_GBS,
          "traffic": traffic,
          "traffic_source": traffic_note,
          "kernel": "orth_project launch sequence of dm_orth_project_f32: k_strip_scatter (the poses "
                    "of the call in its arguments; geometry, projection, owned cells and fill; the cells "
                    "several strips share go to compact planes) + k_strip_combine_planes (one round of "
                    "coalesced loads per shared group) -- everything that produces the "
                    "per-frame maps and masks from the depth maps and the call's poses; the batch "
                    "fuse that follows is excluded here and included in roofline_step",
          "algorithmic_bytes_per_launch": alg,
          "launch_us": kernel_s * 1e6,
          "launch_us_how": b2b_note,
          "launch_us_single_bracketed": bracketed_s * 1e6,
          "launch_us_single_bracketed_note": "one launch sequence between its own pair of events inside the "
                                             "timed steps (includes the events' own stream time)",
          "launch_us_single_bracketed_n": int(len(proj_ms)),
          # what a kernel that only MOVES these bytes reaches on this chip (tools/model.hip `mix`: 78.6 MB of
          # 16-byte loads + 83.9 MB of stores per launch, rotating working set, every block / chunk shape tried:
          # 30.5 us = 5.33 TB/s; reads alone 5.9, writes alone 5.95 -- profiles/r05_traffic_model.log)
          "mix_ceiling_GBps": MIX_CEILING_GBS,
          "frac_of_mix_ceiling": achieved / MIX_CEILING_GBS,
      },
  }
  if step_s is not None:
    alg_step = alg + C_out * mh * mw * 5            # + the batch-fused map and its mask, written once
    result["roofline_step"] = {
        "bound": "hbm", "achieved": alg_step / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": alg_step / step_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_step": alg_step,
        "step_us": step_s * 1e6,
        "how": "the timed loop's own call, orth_project_and_fuse(depth, cam_pose=..., out=..., fused_out=...) "
               "(projection launch sequence + k_fuse_unions), 64 calls back to back between one pair of HIP "
               "events: device time per step, or the host's where the host is the slower one"}

  if not fused_only and C and height_s is not None:
    alg_h = alg + B * mh * mw * 4
    result["roofline_with_height_map"] = {
        "bound": "hbm", "achieved": alg_h / height_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": alg_h / height_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_h,
        "launch_us": height_s * 1e6,
        "how": "orth_project(..., get_height_map=True): the object map + the height map of the same frames "
               "(maps.py:332-350: a second projection, NINF fill, max) -- BASELINE.md's byte count for this config; "
               "8 calls back to back between one pair of HIP events"}
  result["fused_checksum"] = fused_checksum
  if trace is not None:      # host time of every timed step's enqueue, us (DM_BENCH_TRACE=1)
    result["host_step_us"] = [round((t - s0) * 1e6, 1) for s0, t in zip([t0] + trace[:-1], trace)]
    result["device_allocations_in_timed_loop"] = new_segments
  if fused_only:
    split = (ctypes.c_int32 * 4)()
    lib.dm_debug_last_fused_split(split)
    on_strips = lib.dm_debug_last_path() == 2
    result["roofline"]["kernel"] = (
        "dm_orth_project_fused_f32 launch sequence: "
        + (f"k_strip_fused ({split[1]} strips of {split[0]} columns, {split[3]} groups of {split[2]} frame(s)) + k_fuse_windows"
           if on_strips else "k_window_scatter + k_fuse_windows (window path)"))
    result["config"]["workload"] = (f"cfg4: B={B}/GPU, {W}x{H} depth fused straight into one "
                                    f"{mw}x{mh} global map (max)"
                                    f"{' + RCCL all-reduce(max)' if world > 1 else ''}")
  if cfg5:
    result["metric"] = "depth frames/sec projected + ego-flow + cropped, B=16 1280x960->2048x2048"
    result["config"]["workload"] = (
        f"cfg5: B={B}/GPU, {W}x{H} depth -> {mw}x{mh} height maps + the ego-motion flow grid of the same depth "
        f"maps (orth_project_and_flow: one native call), {CFG5_CROP}x{CFG5_CROP} crop of every map + "
        f"mask around its camera's cell")
    result["legs"] = cfg5_legs(dmap, proj, depth_sets, pose_sets, out_sets, centers, flow_tp, fill, rot,
                               back_to_back, B, H, W, mh, mw)
  result["config"]["camera_state"] = (f"a new set of {B} poses on every step, passed per call "
                                      "(MapProjector.orth_project_and_fuse(depth, cam_pose=...)): "
                                      "they travel in the kernel arguments of the call's launches")
  if dist is not None and RING_MAX != 1:
    # the same job with ONE all-reduce per step (DM_BENCH_RING=1 semantics: every step's job-wide fused
    # map is final before the next step is enqueued), so that an N > 1 reader sees what the batching of
    # RING_MAX steps per collective buys and what the strict form costs
    e1, q1, r1, _, _, _ = timed_run(args.steps, min(args.warmup, 4), 1)
    result["ring_of_1"] = {"value": world * B * args.steps / e1, "unit": "frames/s",
                           "ms_per_step": e1 / args.steps * 1e3,
                           "note": "one all-reduce(max) per step instead of one per "
                                   f"{RING_MAX} steps (`value` above): every step's job-wide map is final at once"}
  depth_d, value_d = depth_sets[0], value_sets[0]
  if rank == 0 and world == 1 and not fused_only and not cfg5:
    # the same step on PREPARED frames: one fixed set of poses kept in a device buffer
    # (MapProjector.prepare; what a fixed rig or a captured HIP graph uses) -- same kernels, the
    # poses read from that buffer instead of from the kernel arguments
    try:
      prep = proj.prepare(B, cam_pose=pose_sets[0], value_channels=C)
      n_p = max(20, min(100, args.steps))
      for j in range(5):
        prep.orth_project_and_fuse(depth_sets[j % rot], value_map=value_sets[j % rot], out=out_sets[j % rot],
                                   fused_out=fused_sets[j % rot])
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      for j in range(n_p):
        prep.orth_project_and_fuse(depth_sets[j % rot], value_map=value_sets[j % rot], out=out_sets[j % rot],
                                   fused_out=fused_sets[j % rot])
      torch.cuda.synchronize()
      dt = time.perf_counter() - t1
      us = back_to_back(lambda j: prep.orth_project(depth_sets[j % rot], value_map=value_sets[j % rot],
                                                    out=out_sets[j % rot])) * 1e6
      result["prepared_frames"] = {
          "value": B * n_p / dt, "unit": "frames/s", "steps": n_p, "ms_per_step": dt / n_p * 1e3,
          "launch_us": us,
          "note": "MapProjector.prepare(...).orth_project_and_fuse(depth): one fixed set of poses in a "
                  "device buffer, the same rotating buffers (64 launch sequences back to back between one "
                  "pair of events)"}
      del prep
    except _native.NativeError as e:
      result["prepared_frames"] = {"error": str(e)[:200]}
  if rank == 0 and world == 1 and not fused_only and not cfg5 and rot >= 2 and not args.no_two_streams:
    # the same steps issued alternately on TWO streams (independent batches, their own outputs): reported beside
    # `value`, which stays one stream in program order.  A scatter workgroup fills a CU's LDS, so the second
    # stream's workgroups start where the first's end -- the launches are out of step and the kernel
    # boundaries and heads of one stream run under the loops of the other
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    n_t = max(32, min(128, args.steps)) // 2 * 2

    per_stream = max(1, rot // 2)       # buffer sets of each stream (its own: no two streams write one set)

    def two(n):
      for j in range(n):
        k = (j // 2 % per_stream) * 2 + j % 2          # (rot >= 2: sets 0, 2, ... for stream 0, 1, 3, ... for stream 1)
        with torch.cuda.stream(streams[j % 2]):
          proj.orth_project_and_fuse(depth_sets[k], value_map=value_sets[k],
                                     cam_pose=pose_sets[j % len(pose_sets)], out=out_sets[k],
                                     fused_out=fused_sets[k])
    torch.cuda.synchronize()
    two(8)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    two(n_t)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    result["two_streams"] = {"value": B * n_t / dt, "unit": "frames/s", "steps": n_t, "ms_per_step": dt / n_t * 1e3,
                             "note": "the timed loop's call alternately on two HIP streams (independent batches and "
                                     "output sets; wall clock over all steps): what a caller with two batches in "
                                     "flight gets; NOT the headline `value`"}
  if rank == 0 and world == 1 and args.depth == "uniform" and not fused_only and C == 0 and not cfg5:
    # the same step on scene-like depth (floor + walls: many pixels per cell):
    # reported beside the headline number, outside its timed region
    sdepths = [synthetic_inputs(B, H, W, C, 1234 + rank + 1000 * j, dev, True)[0].to(dev) for j in range(rot)]

    def sstep(j):
      return proj.orth_project_and_fuse(sdepths[j % rot], cam_pose=pose_sets[j % len(pose_sets)],
                                        out=out_sets[j % rot], fused_out=fused_sets[j % rot])
    for j in range(10):
      sstep(j)
    torch.cuda.synchronize()
    n_s = 100
    t1 = time.perf_counter()
    for j in range(n_s):
      sstep(j)
    torch.cuda.synchronize()
    result["scene_like_depth"] = {"value": B * n_s / (time.perf_counter() - t1), "unit": "frames/s",
                                  "steps": n_s, "note": "same workload on floor + walls depth"}
    del sdepths
  if rank == 0 and world == 1 and not args.no_cpu_baseline and not fused_only:
    # the last timed step's poses on depth batch 0 (the host copy the oracle reads) into fresh
    # tensors: the rotating output sets have been overwritten by the measurements above
    out = proj.orth_project_and_fuse(depth_sets[0], value_map=value_sets[0], cam_pose=last_pose)
    result["cpu_baseline"] = cpu_baseline(depth, last_pose, value, H, W, mh, mw, fill,
                                          args.cpu_seconds, out)
  if rank == 0 and world == 1 and not args.no_other_configs and args.workload == "cfg2" \
      and args.depth == "uniform":
    del depth_sets, value_sets, out_sets, fused_sets, out
    torch.cuda.empty_cache()
    result["other_configs"] = other_configs(dmap, lib, dev)
  if rank == 0:
    print(json.dumps(result))
  if dist is not None:
    dist.destroy_process_group()


def cfg5_sample_checks(dmap, proj, depth, pose, flow_tp, center, top, mask, fill, H, W, mh, mw):
  """Frame 0 of a cfg5 batch against the oracle: the ego-motion flow grid (oracle.camera_affine_grid), the
  grid of the fused projection + flow call, and the crop of its map + mask around the camera's cell
  (oracle.crop_nearest) -- bit for bit.  depth (1, 1, H, W) device, top / mask (1, 1, mh, mw) device."""
  from oracle import oracle
  cx, cy, fx, fy = oracle.camera_intrinsics(W, H, np.radians(70.))
  want_grid = oracle.camera_affine_grid(depth.cpu().numpy(), flow_tp.numpy(), np.radians(-20.), 0.88, fx, fy, cx, cy, True)
  grid = proj.camera_affine_grid(depth, flow_tp).cpu().numpy()
  same = lambda a, b: bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())
  t2, m2, g2 = proj.orth_project_and_flow(depth, flow_tp, cam_pose=pose)
  from dungeon_maps_amd import _native
  _native.lib().dm_debug_flow_fused(1)
  try:
    t3, m3, g3 = proj.orth_project_and_flow(depth, flow_tp, cam_pose=pose)
  finally:
    _native.lib().dm_debug_flow_fused(0)
  crop, crop_mask = dmap.functional.crop_nearest(top, center, CFG5_CROP, CFG5_CROP, fill_value=fill, mask=mask)
  want_crop = oracle.crop_nearest(top.cpu().numpy(), center.cpu().numpy(), CFG5_CROP, CFG5_CROP, fill)
  want_cmask = oracle.crop_nearest(mask.cpu().numpy(), center.cpu().numpy(), CFG5_CROP, CFG5_CROP, False)
  return {"camera_affine_grid": same(grid, want_grid),
          "orth_project_and_flow": same(g2.cpu().numpy(), want_grid) and bool(torch.equal(t2, top) and torch.equal(m2, mask)),
          "orth_project_and_flow_one_kernel": same(g3.cpu().numpy(), want_grid) and bool(torch.equal(t3, top) and torch.equal(m3, mask)),
          "crop_topdown_map": same(crop.cpu().numpy(), want_crop) and bool(np.array_equal(crop_mask.cpu().numpy(), want_cmask))}


def cfg5_legs(dmap, proj, depth_sets, pose_sets, out_sets, centers, flow_tp, fill, rot, back_to_back,
              B, H, W, mh, mw):
  """The three legs of the cfg5 step on their own (64 calls back to back each, rotating buffers):
  device time, algorithmic bytes and fraction of the HBM peak; frame 0 of every leg is
  checked against the oracle."""
  legs = {}

  def leg(name, fn, alg, note):
    us = back_to_back(fn, 32) * 1e6
    legs[name] = {"us": us, "algorithmic_bytes": alg, "achieved_GBps": alg / us / 1e3,
                  "frac": alg / us / 1e3 / HBM_PEAK_GBS, "what": note}
  keep = [None] * (rot + 1)

  def plain(j):
    keep[j % len(keep)] = None
    keep[j % len(keep)] = proj.orth_project(depth_sets[j % rot], cam_pose=pose_sets[j % len(pose_sets)])
  leg("orth_project", plain, algorithmic_bytes(B, H, W, mh, mw, 0),
      "per-frame height maps + masks (maps.py:127-351)")
  both = [None] * 2

  def fused_flow(j):
    both[j % 2] = None
    both[j % 2] = proj.orth_project_and_flow(depth_sets[j % rot], flow_tp, cam_pose=pose_sets[j % len(pose_sets)],
                                            out=out_sets[j % rot])
  leg("orth_project_and_flow", fused_flow, algorithmic_bytes(B, H, W, mh, mw, 0) + B * H * W * (4 + 8),
      "both from one native call (dm_orth_project_flow_f32), as the timed steps make it: the projection's "
      "launches and the flow kernel behind them (depth read twice)")
  from dungeon_maps_amd import _native
  _native.lib().dm_debug_flow_fused(1)
  try:
    leg("orth_project_and_flow_one_kernel", fused_flow, algorithmic_bytes(B, H, W, mh, mw, 0) + B * H * W * 8,
        "the same call with the projection kernel computing the flow from the depth it has loaded (one depth "
        "read; dm_debug_flow_fused(1)): slower than the two kernels on this chip, hence not the default")
  finally:
    _native.lib().dm_debug_flow_fused(0)
  del both
  grids = [None] * 2

  def flow(j):
    grids[j % 2] = None
    grids[j % 2] = proj.camera_affine_grid(depth_sets[j % rot], flow_tp)
  leg("camera_affine_grid", flow, B * H * W * (4 + 8),
      "ego-motion flow grid of the same depth maps (maps.py:353-460): depth in, two floats per pixel out")
  crops = [None] * 2

  def crop(j):
    crops[j % 2] = None
    crops[j % 2] = dmap.functional.crop_nearest(out_sets[j % rot][0], centers[j % len(centers)], CFG5_CROP,
                                                CFG5_CROP, fill_value=fill, mask=out_sets[j % rot][1])
  leg("crop_topdown_map", crop, 2 * B * CFG5_CROP * CFG5_CROP * 5,
      f"dm_crop_nearest_f32: {CFG5_CROP}x{CFG5_CROP} crop of every height map + mask around its camera's "
      "cell (maps.py:1959-2037, utils.py:571-652): 5 bytes per cell in, 5 out")
  # frame 0 of every leg against the oracle (the projection itself: orth_project on frame 0)
  from oracle import oracle
  d0, p0 = depth_sets[0][:1], pose_sets[0][:1]
  top0, mask0 = proj.orth_project(d0, cam_pose=p0)
  cx, cy, fx, fy = oracle.camera_intrinsics(W, H, np.radians(70.))
  want = oracle.orth_project(d0.cpu().numpy(), cam_pose=p0.numpy(), width_offset=mw / 2., height_offset=mh / 2.,
                             cam_pitch=np.radians(-20.), cam_height=0.88, map_res=0.03, map_width=mw, map_height=mh,
                             focal_x=fx, focal_y=fy, center_x=cx, center_y=cy, trunc_depth_min=0.15,
                             trunc_depth_max=5.05, to_global=True, fill_value=fill)
  checks = cfg5_sample_checks(dmap, proj, d0, p0, flow_tp, centers[0][:1], top0, mask0, fill, H, W, mh, mw)
  checks["orth_project"] = bool(np.array_equal(top0.cpu().numpy(), want[0]) and np.array_equal(mask0.cpu().numpy(), want[1]))
  for name, ok in checks.items():
    if name in legs:
      legs[name]["gpu_matches_cpu_on_sample"] = ok
  return legs


def committed_traffic(workload, lib_path):
  """(bytes per launch sequence, note) from profiles/r05_hbm_traffic.json if its `lib_md5`
  is the md5 of the loaded library, else (None, why)."""
  import hashlib
  tpath = os.path.join(ROOT, "profiles", "r05_hbm_traffic.json")
  if workload != "cfg2" or not os.path.exists(tpath):
    return None, "no committed PMC summary for this workload"
  with open(tpath) as f:
    rec = json.load(f)
  with open(lib_path, "rb") as f:
    md5 = hashlib.md5(f.read()).hexdigest()
  if rec.get("lib_md5") != md5:
    return None, (f"profiles/r05_hbm_traffic.json was measured on library build "
                  f"{rec.get('lib_md5')}, the loaded one is {md5}: not quoted")
  return rec.get("launch_sequence_bytes"), (
      "profiles/r05_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command on "
      "this library build, gfx950 correction applied; bytes per launch sequence)")


def _event_us(fn, lib, n, hooks=True):
  """Mean HIP-event time (us) of fn()'s projection sequence over n calls, and the wall time
  per call of a back-to-back loop."""
  ea = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
  eb = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
  for e in ea + eb:
    e.record()
  for _ in range(3):
    fn()
  torch.cuda.synchronize()
  for i in range(n):
    if hooks:
      lib.dm_debug_record_before_projection(ea[i].cuda_event)
      lib.dm_debug_record_after_projection(eb[i].cuda_event)
      fn()
    else:
      fn()
  torch.cuda.synchronize()
  if hooks:
    ev = float(np.mean([ea[i].elapsed_time(eb[i]) for i in range(n)])) * 1e3
  else:
    # (no hook inside this entry point: n calls back to back between ONE pair of events -- events
    # around every single call would time the host's part of the call as well)
    ea[0].record()
    for _ in range(n):
      fn()
    eb[0].record()
    torch.cuda.synchronize()
    ev = ea[0].elapsed_time(eb[0]) * 1e3 / n
  t0 = time.perf_counter()
  for _ in range(n):
    fn()
  torch.cuda.synchronize()
  return ev, (time.perf_counter() - t0) / n * 1e6


def other_configs(dmap, lib, dev):
  """The other BASELINE.json configs, outside the headline's timed region: HIP-event time of the
  projection sequence, algorithmic bytes, fraction of the 8 TB/s HBM peak, and a
  check against the CPU oracle on a bounded sample.  Kept short (a few seconds in all)."""
  from oracle import oracle
  res = {}
  g = torch.Generator(device=dev).manual_seed(4242)
  cxyf = lambda W, H: oracle.camera_intrinsics(W, H, np.radians(70.))

  def projector(H, W, mh, mw, fill):
    return dmap.MapProjector(
        width=W, height=H, hfov=np.radians(70.), cam_pitch=np.radians(-20.), cam_height=0.88,
        width_offset=mw / 2., height_offset=mh / 2., map_res=0.03, map_width=mw, map_height=mh,
        trunc_depth_min=0.15, trunc_depth_max=5.05, clip_border=0, to_global=True,
        fill_value=fill, reduction="max")

  def okw(H, W, mh, mw, fill):
    cx, cy, fx, fy = cxyf(W, H)
    return dict(width_offset=mw / 2., height_offset=mh / 2., cam_pitch=np.radians(-20.),
                cam_height=0.88, map_res=0.03, map_width=mw, map_height=mh, focal_x=fx, focal_y=fy,
                center_x=cx, center_y=cy, trunc_depth_min=0.15, trunc_depth_max=5.05,
                to_global=True, fill_value=fill)

  def poses(B):
    p = torch.empty(B, 3, device=dev).uniform_(-1, 1, generator=g)
    p[:, 2] = torch.empty(B, device=dev).uniform_(-np.pi, np.pi, generator=g)
    return p.cpu()

  def entry(name, us, wall_us, alg, same, note, cpu=None):
    res[name] = {"launch_us": us, "call_wall_us": wall_us, "algorithmic_bytes": alg,
                 "achieved_GBps": alg / us / 1e3, "frac": alg / us / 1e3 / HBM_PEAK_GBS,
                 "gpu_matches_cpu_on_sample": same, "workload": note}
    if cpu is not None:
      res[name]["cpu_baseline"] = cpu

  try:
    avail = len(os.sched_getaffinity(0))
  except AttributeError:
    avail = os.cpu_count() or 1
  cores = max(1, min(oracle.max_threads(), avail))

  def cpu_rate(fn, frames, threads, what, budget_s=2.0):
    """The oracle (a port of the reference's algorithm) on a bounded sample of the config, frames/s: median of
    the runs that fit the budget (at least two)."""
    times = []
    t_end = time.perf_counter() + budget_s
    while time.perf_counter() < t_end or len(times) < 2:
      t0 = time.perf_counter()
      fn()
      times.append(time.perf_counter() - t0)
    return {"value": frames / float(np.median(times)), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{what}; median of {len(times)} runs (~{sum(times):.1f} s)"}

  # cfg1: B=1, 320x240 -> 256x256 (the reference's own demo size)
  B, H, W, mh, mw, _ = WORKLOADS["cfg1"]
  d = torch.empty(B, 1, H, W, device=dev).uniform_(0.1, 10.0, generator=g)
  po = poses(B)
  proj = projector(H, W, mh, mw, -np.inf)
  us, wall = _event_us(lambda: proj.orth_project(d, cam_pose=po), lib, 50)
  top, mask = proj.orth_project(d, cam_pose=po)
  want = oracle.orth_project(d.cpu().numpy(), cam_pose=po.numpy(), **okw(H, W, mh, mw, -np.inf))
  same = bool(np.array_equal(top.cpu().numpy(), want[0]) and np.array_equal(mask.cpu().numpy(), want[1]))
  d_np, po_np = d.cpu().numpy(), po.numpy()
  entry("cfg1", us, wall, algorithmic_bytes(B, H, W, mh, mw, 0), same,
        "B=1, 320x240 -> 256x256 height map, orth_project per call",
        cpu_rate(lambda: oracle.orth_project(d_np, cam_pose=po_np, **okw(H, W, mh, mw, -np.inf)), 1, 1,
                 "the config's one frame on one thread"))
  # the same frame with the camera state prepared once and the call replayed from a HIP graph (what a
  # fixed-rig loop can do: no host geometry, no copy, one graph launch per frame); events around
  # 200 back-to-back replays, total / 200
  try:
    prep1 = proj.prepare(B, cam_pose=po)
    outs = (torch.empty((B, 1, mh, mw), dtype=torch.float32, device=dev),
            torch.empty((B, 1, mh, mw), dtype=torch.bool, device=dev))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(3):
        prep1.orth_project(d, out=outs)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      prep1.orth_project(d, out=outs)
    for _ in range(5):
      graph.replay()
    torch.cuda.synchronize()
    n = 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
      graph.replay()
    e1.record()
    torch.cuda.synchronize()
    wall_g = (time.perf_counter() - t0) / n * 1e6
    us_g = e0.elapsed_time(e1) * 1e3 / n
    same_g = bool(np.array_equal(outs[0].cpu().numpy(), want[0]) and np.array_equal(outs[1].cpu().numpy(), want[1]))
    entry("cfg1_prepared_graph", us_g, wall_g, algorithmic_bytes(B, H, W, mh, mw, 0), same_g,
          "the same frame through MapProjector.prepare + a captured HIP graph, per replay "
          "(200 replays back to back)")
    del graph, prep1
  except Exception as e:      # (a library that cannot prepare this shape: reported, not fatal)
    res["cfg1_prepared_graph"] = {"error": str(e)[:200]}

  def spin(n_sets, make):
    """`n_sets` input sets in rotation and the last results kept alive (so that the allocator hands out
    other output blocks call after call): what a call touches comes around after more than the
    Infinity Cache holds, as in the headline's loop."""
    sets = [make(j) for j in range(n_sets)]
    keep, count = [None] * (n_sets + 1), [0]

    def fn_of(call):
      def fn():
        j = count[0]
        count[0] = j + 1
        keep[j % len(keep)] = None
        keep[j % len(keep)] = call(sets[j % n_sets])
      return fn
    return sets, fn_of

  def rot_of(alg):
    return max(1, min(8, -(-ROT_TARGET_BYTES // alg) + 1))

  # cfg3: B=64, 40-class one-hot object map (no height map), checked on frame 0
  B, H, W, mh, mw, C = WORKLOADS["cfg3"]
  po = poses(B)

  def make3(j):
    d = torch.empty(B, 1, H, W, device=dev).uniform_(0.1, 10.0, generator=g)
    labels = torch.randint(0, C, (B, H, W), device=dev, generator=g)
    v = torch.zeros(B, C, H, W, device=dev)
    v.scatter_(1, labels.unsqueeze(1), 1.0)
    return d, v
  alg3 = algorithmic_bytes(B, H, W, mh, mw, C)
  proj = projector(H, W, mh, mw, 0.0)
  sets, fn_of = spin(rot_of(alg3), make3)
  us, wall = _event_us(fn_of(lambda s_: proj.orth_project(s_[0], value_map=s_[1], cam_pose=po)), lib, 4)
  d, v = sets[0]
  top, mask = proj.orth_project(d, value_map=v, cam_pose=po)
  want = oracle.orth_project(d[:1].cpu().numpy(), value_map=v[:1].cpu().numpy(),
                             cam_pose=po[:1].numpy(), **okw(H, W, mh, mw, 0.0))
  same = bool(np.array_equal(top[:1].cpu().numpy(), want[0]) and np.array_equal(mask[:1].cpu().numpy(), want[1]))
  n3 = min(B, max(8, cores))
  d_np, v_np, po_np = d[:n3].cpu().numpy(), v[:n3].cpu().numpy(), po[:n3].numpy()
  cpu3 = cpu_rate(lambda: oracle.orth_project(d_np, value_map=v_np, cam_pose=po_np, nthreads=min(cores, n3),
                                              **okw(H, W, mh, mw, 0.0)), n3, min(cores, n3),
                  f"{n3} frames of the batch, one OpenMP thread per frame", 3.0)
  entry("cfg3", us, wall, alg3, same,
        "B=64, 640x480 + 40-class one-hot -> 512x512 object map, no height map (checked on frame 0)", cpu3)
  # ... and as BASELINE.md section 4 counts it: WITH the height map (maps.py:332-350: the second
  # projection MapBuilder.plot always asks for, maps.py:2448-2455), + B * mh * mw * 4 bytes written
  del top, mask
  us_h, wall_h = _event_us(fn_of(lambda s_: proj.orth_project(s_[0], value_map=s_[1], cam_pose=po, get_height_map=True)),
                           lib, 4)
  top, mask, hmap = proj.orth_project(d, value_map=v, cam_pose=po, get_height_map=True)
  want = oracle.orth_project(d[:1].cpu().numpy(), value_map=v[:1].cpu().numpy(), cam_pose=po[:1].numpy(),
                             get_height_map=True, **okw(H, W, mh, mw, 0.0))
  same_h = bool(np.array_equal(top[:1].cpu().numpy(), want[0]) and np.array_equal(mask[:1].cpu().numpy(), want[1])
                and np.array_equal(hmap[:1, :1].cpu().numpy(), np.asarray(want[2])[:, :1]))
  entry("cfg3_with_height_map", us_h, wall_h, alg3 + B * mh * mw * 4, same_h,
        "the same + get_height_map=True (BASELINE.md's 6 646 923 264 B per batch: what MapBuilder.plot asks for; "
        "object map, mask and height map of frame 0 checked)")
  del v, top, mask, hmap, sets, fn_of, d
  torch.cuda.empty_cache()

  # cfg4 per rank: 64 frames of one trajectory fused straight into ONE 1024x1024 map
  B, H, W, mh, mw, _ = WORKLOADS["cfg4"]
  k = torch.arange(B, dtype=torch.float32)
  po = torch.stack((0.02 * k, 0.01 * k, 0.01 * k), dim=1)
  proj = projector(H, W, mh, mw, -np.inf)
  alg4 = algorithmic_bytes(B, H, W, mh, mw, 0, True)
  sets, fn_of = spin(rot_of(alg4), lambda j: torch.empty(B, 1, H, W, device=dev).uniform_(0.1, 10.0, generator=g))
  us, wall = _event_us(fn_of(lambda d_: proj.orth_project_fused(d_, cam_pose=po)), lib, 20, hooks=False)
  d = sets[0]
  fused, fmask = proj.orth_project_fused(d, cam_pose=po)
  want = oracle.orth_project(d.cpu().numpy(), cam_pose=po.numpy(), fused=True, **okw(H, W, mh, mw, -np.inf))
  same = bool(np.array_equal(fused.cpu().numpy(), want[0]) and np.array_equal(fmask.cpu().numpy(), want[1]))
  d_np, po_np = d[:8].cpu().numpy(), po[:8].numpy()
  entry("cfg4_per_gpu", us, wall, alg4, same,
        "64 frames/GPU of one trajectory fused into one 1024x1024 map (20 calls back to back between one pair of events: the larger of the device's and the host's time per call)",
        cpu_rate(lambda: oracle.orth_project(d_np, cam_pose=po_np, fused=True, **okw(H, W, mh, mw, -np.inf)), 8, 1,
                 "the trajectory's first 8 frames fused into the one map, one thread (the fused port is serial)"))
  del sets, fn_of, d
  torch.cuda.empty_cache()

  # cfg5 per GPU: 16 frames of 1280x960 -> 2048x2048, plus the ego-motion flow grid
  B, H, W, mh, mw = 16, 960, 1280, 2048, 2048
  po = poses(B)
  proj = projector(H, W, mh, mw, -np.inf)
  alg5 = algorithmic_bytes(B, H, W, mh, mw, 0)
  sets, fn_of = spin(rot_of(alg5), lambda j: torch.empty(B, 1, H, W, device=dev).uniform_(0.1, 10.0, generator=g))
  us, wall = _event_us(fn_of(lambda d_: proj.orth_project(d_, cam_pose=po)), lib, 6)
  d = sets[0]
  top, mask = proj.orth_project(d, cam_pose=po)
  want = oracle.orth_project(d[:1].cpu().numpy(), cam_pose=po[:1].numpy(), **okw(H, W, mh, mw, -np.inf))
  same = bool(np.array_equal(top[:1].cpu().numpy(), want[0]) and np.array_equal(mask[:1].cpu().numpy(), want[1]))
  n5 = min(B, max(4, min(cores, 8)))
  d_np, po_np = d[:n5].cpu().numpy(), po[:n5].numpy()
  entry("cfg5_per_gpu", us, wall, alg5, same,
        "16 frames/GPU, 1280x960 -> 2048x2048 height map (checked on frame 0)",
        cpu_rate(lambda: oracle.orth_project(d_np, cam_pose=po_np, nthreads=min(cores, n5), **okw(H, W, mh, mw, -np.inf)),
                 n5, min(cores, n5), f"{n5} frames of the batch, one OpenMP thread per frame"))
  tp = torch.tensor([0.05, 0.1, 0.02])
  center0 = torch.stack((po[:1, 0] / 0.03 + mw / 2., (mh - 1) - (po[:1, 1] / 0.03 + mh / 2.)), dim=1).to(dev)
  checks5 = cfg5_sample_checks(dmap, proj, d[:1], po[:1], tp, center0, top[:1], mask[:1], -np.inf, H, W, mh, mw)
  centers5 = torch.stack((po[:, 0] / 0.03 + mw / 2., (mh - 1) - (po[:, 1] / 0.03 + mh / 2.)), dim=1).to(dev)
  crops = [None] * 2
  ncrop = [0]

  def crop_call():
    j = ncrop[0]
    ncrop[0] += 1
    crops[j % 2] = None
    crops[j % 2] = dmap.functional.crop_nearest(top, centers5, CFG5_CROP, CFG5_CROP, fill_value=-np.inf, mask=mask)
  us_c, wall_c = _event_us(crop_call, lib, 6, hooks=False)
  alg_c = 2 * B * CFG5_CROP * CFG5_CROP * 5
  res["cfg5_crop"] = {"launch_us": us_c, "call_wall_us": wall_c, "algorithmic_bytes": alg_c,
                      "achieved_GBps": alg_c / us_c / 1e3, "frac": alg_c / us_c / 1e3 / HBM_PEAK_GBS,
                      "gpu_matches_cpu_on_sample": checks5["crop_topdown_map"],
                      "workload": f"{CFG5_CROP}x{CFG5_CROP} crop of the 16 maps + masks around their cameras' cells "
                                  "(TopdownMap.select; frame 0 checked against oracle.crop_nearest)"}
  del crops, top, mask
  us, wall = _event_us(fn_of(lambda d_: proj.camera_affine_grid(d_, tp)), lib, 6, hooks=False)
  alg = B * H * W * (4 + 8)
  res["cfg5_ego_flow"] = {"launch_us": us, "call_wall_us": wall, "algorithmic_bytes": alg,
                          "achieved_GBps": alg / us / 1e3, "frac": alg / us / 1e3 / HBM_PEAK_GBS,
                          "gpu_matches_cpu_on_sample": checks5["camera_affine_grid"],
                          "fused_call_matches_cpu_on_sample": checks5["orth_project_and_flow"],
                          "workload": "camera_affine_grid of the same 16 frames (6 calls back to back between one pair "
                                      "of events; frame 0 checked against oracle.camera_affine_grid, as is the grid "
                                      "of the fused orth_project_and_flow call)"}
  res["working_sets"] = ("cfg3, cfg4 and cfg5 rotate over input batches and output blocks like the headline loop (what a call "
                         "touches comes around after > 2 x 256 MiB of other traffic); cfg1 is one 0.6 MB frame")
  return res


def cpu_baseline(depth, pose, value, H, W, mh, mw, fill, budget_s, gpu_out):
  """The CPU oracle (port of the reference's algorithm, oracle/dm_oracle.c, one OpenMP
  thread per frame) on a bounded sample of the same workload, on the host cores this
  process may use; also the last check of the GPU result.  Context, never the target."""
  from oracle import oracle
  try:
    avail = len(os.sched_getaffinity(0))
  except AttributeError:
    avail = os.cpu_count() or 1
  cores = max(1, min(oracle.max_threads(), avail))
  B = depth.shape[0]
  # the oracle parallelises over frames: the sample holds at least as many frames as there
  # are cores (the batch repeated), so every core has work
  reps = 1 if value is not None else max(1, min(4, -(-cores // B)))
  n = B * reps if value is None else min(B, max(cores, 8))
  d = depth[:B].numpy() if reps == 1 else np.concatenate([depth.numpy()] * reps, 0)
  d = d[:n]
  po = (pose.numpy() if reps == 1 else np.concatenate([pose.numpy()] * reps, 0))[:n]
  v = None if value is None else value[:n].numpy()
  threads = min(cores, n)
  cx, cy, fx, fy = oracle.camera_intrinsics(W, H, np.radians(70.))
  kw = dict(width_offset=mw / 2., height_offset=mh / 2.,
            cam_pitch=np.radians(-20.), cam_height=0.88, map_res=0.03, map_width=mw,
            map_height=mh, focal_x=fx, focal_y=fy, center_x=cx, center_y=cy,
            trunc_depth_min=0.15, trunc_depth_max=5.05, to_global=True, fill_value=fill)
  want = oracle.orth_project(d, value_map=v, cam_pose=po, nthreads=threads, **kw)   # warm-up + parity
  m = min(n, B)
  same = bool(np.array_equal(gpu_out[0][:m].cpu().numpy(), want[0][:m])
              and np.array_equal(gpu_out[1][:m].cpu().numpy(), want[1][:m]))
  times = []
  t_end = time.perf_counter() + budget_s
  while time.perf_counter() < t_end or len(times) < 3:
    t0 = time.perf_counter()
    oracle.orth_project(d, value_map=v, cam_pose=po, nthreads=threads, **kw)
    times.append(time.perf_counter() - t0)
  k1 = 2 if value is None else 1                   # one thread, a couple of frames
  t1 = []
  for _ in range(3):
    t0 = time.perf_counter()
    oracle.orth_project(d[:k1], value_map=None if v is None else v[:k1], cam_pose=po[:k1],
                        nthreads=1, **kw)
    t1.append(time.perf_counter() - t0)
  return {
      "value": n / float(np.median(times)),
      "unit": "frames/s",
      "cores": threads,
      "cores_available": avail,
      "single_thread_value": k1 / float(np.median(t1)),
      "kind": "port",
      "sample": f"{n} frames of the same workload ({'the batch repeated ' + str(reps) + 'x' if reps > 1 else 'a slice of the batch'}), "
                f"median of {len(times)} runs (~{sum(times):.0f} s), one OpenMP thread per frame, "
                f"{threads} threads; single_thread_value = {k1} frame(s) on one thread",
      "gpu_matches_cpu_on_sample": same,
  }


if __name__ == "__main__":
  main()
