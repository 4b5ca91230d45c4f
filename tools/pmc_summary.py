"""Per-kernel PMC summaries of bench.py runs under rocprofv3 (one counter set per pass, as
MI355X_MICROARCH.md prescribes):

    python tools/pmc_summary.py <session dir> <workload> <out prefix>

reads <dir>/fetch_<workload>, <dir>/write_<workload> (FETCH_SIZE / WRITE_SIZE passes), <dir>/sq_<workload>
(SQ counters) and <dir>/stats_<workload> (--kernel-trace --stats), and writes <out prefix>_<workload>_hbm_traffic.json
(quoted by bench.py as roofline.traffic when its lib_md5 is the loaded library's) plus a markdown table on stdout.
Units: FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads and
is doubled, WRITE_SIZE is taken as read; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("k_strip_scatter", "k_strip_combine_planes", "k_strip_combine", "k_fuse_unions", "k_strip_fused", "k_fuse_windows", "k_window_scatter",
           "k_window_merge", "k_camera_affine_grid", "k_crop_nearest")
# (exact names: at cfg3 the height pass of the with-height-map leg runs k_strip_scatter + k_strip_combine_planes too)
SEQUENCE = {"cfg2": ("k_strip_scatter", "k_strip_combine_planes"),
            "cfg3": ("k_strip_scatter (index pass)", "k_strip_scatter (value pass)", "k_strip_combine"),
            "cfg4": ("k_strip_fused", "k_fuse_windows"), "cfg5": ("k_window_scatter", "k_window_merge")}
ALG = {"cfg2": 64 * (480 * 640 * 4 + 512 * 512 * 5), "cfg3": 64 * (480 * 640 * 4 * 41 + 40 * 512 * 512 * 5),
       "cfg4": 64 * 480 * 640 * 4 + 1024 * 1024 * 5, "cfg5": 16 * (960 * 1280 * 4 + 2048 * 2048 * 5)}


def short(name):
  for k in KERNELS:
    if k in name:
      if k == "k_strip_scatter":      # the pass: whole projection / index pass / value pass (template parameter MODE)
        args = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",")
        mode = args[4] if len(args) > 4 else "0"
        return {"0": "k_strip_scatter", "1": "k_strip_scatter (index pass)", "2": "k_strip_scatter (value pass)"}.get(mode, k)
      return k
  return None


def counters(directory):
  acc = defaultdict(lambda: defaultdict(list))
  for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
      for row in csv.DictReader(f):
        k = short(row["Kernel_Name"])
        if k:
          acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
  return {k: {c: (sum(v) / len(v), len(v)) for c, v in d.items()} for k, d in acc.items()}


def durations(directory):
  out = {}
  for path in glob.glob(os.path.join(directory, "**", "*kernel_stats.csv"), recursive=True):
    with open(path) as f:
      for row in csv.DictReader(f):
        k = short(row["Name"])
        if k:
          calls, avg = int(row["Calls"]), float(row["AverageNs"])
          if k in out:      # several template instances of one kernel: weighted by calls
            c0, a0 = out[k]
            out[k] = (c0 + calls, (a0 * c0 + avg * calls) / (c0 + calls))
          else:
            out[k] = (calls, avg)
  return out


def main():
  d, w, prefix = sys.argv[1:4]
  fetch = counters(os.path.join(d, f"fetch_{w}"))
  write = counters(os.path.join(d, f"write_{w}"))
  sq = counters(os.path.join(d, f"sq_{w}"))
  dur = durations(os.path.join(d, f"stats_{w}"))
  lib = os.path.join(ROOT, "dungeon_maps_amd", "csrc", "libdungeon_maps_amd.so")
  with open(lib, "rb") as f:
    md5 = hashlib.md5(f.read()).hexdigest()
  kernels = {}
  for k in sorted(set(fetch) | set(write)):
    if k in fetch and k in write and "FETCH_SIZE" in fetch[k] and "WRITE_SIZE" in write[k]:
      kernels[k] = {"read": int(round(fetch[k]["FETCH_SIZE"][0] * 1024 * 2)),
                    "written": int(round(write[k]["WRITE_SIZE"][0] * 1024)),
                    "dispatches": fetch[k]["FETCH_SIZE"][1]}
      if k in dur:
        kernels[k]["avg_us"] = dur[k][1] / 1e3
  seq_names = [k for k in kernels if k in SEQUENCE[w]]
  # per launch sequence: a kernel that runs n times per call (channel groups) counts n times
  calls = min((kernels[k]["dispatches"] for k in seq_names), default=1)
  seq = sum((kernels[k]["read"] + kernels[k]["written"]) * kernels[k]["dispatches"] / calls for k in seq_names)
  rec = {"command": f"rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py "
                    f"--gpus 1 --workload {w} ... (one counter per pass; tools/sessions/)",
         "workload": w, "lib_md5": md5, "unit": "bytes per dispatch (averages over the dispatches of the run)",
         "correction": "both counters in KB; FETCH_SIZE doubled (gfx950 tallies the 128-B requests of wide coalesced "
                       "reads at 64 B, MI355X_MICROARCH.md); WRITE_SIZE as read.  Requests the Infinity Cache serves are "
                       "counted too; the benchmark rotates its buffers so that none of a step's bytes can be resident",
         "kernels": kernels, "launch_sequence_bytes": int(seq), "algorithmic_bytes": ALG[w], "ratio": seq / ALG[w]}
  with open(f"{prefix}_{w}_hbm_traffic.json" if w != "cfg2" else f"{prefix}_hbm_traffic.json", "w") as f:
    json.dump(rec, f, indent=1)
  print(f"## {w}: library {md5}\n")
  print("| kernel | avg us (rocprofv3 --stats) | read MB | written MB | dispatches |\n|---|---|---|---|---|")
  for k, v in kernels.items():
    print(f"| {k} | {v.get('avg_us', float('nan')):.1f} | {v['read'] / 1e6:.1f} | {v['written'] / 1e6:.1f} | {v['dispatches']} |")
  print(f"\nLaunch sequence ({' + '.join(seq_names)}): **{seq / 1e6:.1f} MB** against **{ALG[w] / 1e6:.1f} MB** algorithmic "
        f"({seq / ALG[w]:.2f}x).\n")
  if sq:
    print("| kernel | waves | VALU instructions / wave | SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY | "
          "VALU issue time per SIMD (4 cycles per instruction, 1024 SIMDs) |\n|---|---|---|---|---|---|---|---|")
    for k, c in sq.items():
      g = lambda n: c.get(n, (float("nan"), 0))[0]
      waves, insts, wc = g("SQ_WAVES"), g("SQ_INSTS_VALU"), g("SQ_WAVE_CYCLES")
      print(f"| {k} | {waves:.0f} | {insts / waves:.0f} | {g('SQ_ACTIVE_INST_VALU') / wc:.2f} | {g('SQ_WAIT_ANY') / wc:.2f} | "
            f"{g('SQ_WAIT_INST_ANY') / wc:.2f} | {g('SQ_ACTIVE_INST_ANY') / wc:.2f} | {insts / 1024 * 4:.0f} cycles |")
    print()


if __name__ == "__main__":
  main()
