"""HBM traffic of the cfg2 launch sequence from two rocprofv3 PMC passes (one counter per pass, as
MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- \
        python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- \
        python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_hbm_traffic

Writes <out>.json (which bench.py quotes as roofline.traffic when its lib_md5 is the md5 of the
loaded library) and <out>.md.  Units: both counters are in KB; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads and is doubled, WRITE_SIZE is taken as read."""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEQUENCE = ("k_strip_scatter", "k_strip_combine")          # the orth_project launch sequence
OTHERS = ("k_fuse_unions",)


def per_kernel(directory, counter):
  acc = defaultdict(list)
  for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
      for row in csv.DictReader(f):
        if row["Counter_Name"] != counter:
          continue
        name = row["Kernel_Name"]
        for key in SEQUENCE + OTHERS:
          if key in name:
            acc[key].append(float(row["Counter_Value"]))
  return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
  fetch_dir, write_dir, out = sys.argv[1:4]
  fetch = per_kernel(fetch_dir, "FETCH_SIZE")
  write = per_kernel(write_dir, "WRITE_SIZE")
  lib = os.path.join(ROOT, "dungeon_maps_amd", "csrc", "libdungeon_maps_amd.so")
  with open(lib, "rb") as f:
    md5 = hashlib.md5(f.read()).hexdigest()
  kernels = {}
  for k in SEQUENCE + OTHERS:
    if k in fetch and k in write:
      kernels[k] = {"read": int(round(fetch[k][0] * 1024 * 2)), "written": int(round(write[k][0] * 1024)),
                    "dispatches": fetch[k][1]}
  seq = sum(kernels[k]["read"] + kernels[k]["written"] for k in SEQUENCE if k in kernels)
  alg = 64 * (480 * 640 * 4 + 512 * 512 * 5)
  rec = {
      "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py "
                 "--steps 5 --warmup 2 --no-cpu-baseline --no-other-configs (one counter per pass)",
      "workload": "cfg2",
      "lib_md5": md5,
      "unit": "bytes per launch (averages over the dispatches of the run)",
      "correction": "both counters in KB; FETCH_SIZE doubled (gfx950 tallies the 128-B requests of wide "
                    "coalesced reads at 64 B, MI355X_MICROARCH.md); WRITE_SIZE as read",
      "kernels": kernels,
      "launch_sequence_bytes": seq,
      "algorithmic_bytes": alg,
      "ratio": seq / alg,
  }
  with open(out + ".json", "w") as f:
    json.dump(rec, f, indent=1)
  with open(out + ".md", "w") as f:
    f.write("# Round 3 -- HBM traffic of the cfg2 launch sequence (PMC, MI355X)\n\n")
    f.write("    " + rec["command"] + "\n\n")
    f.write("Library build (md5 of libdungeon_maps_amd.so): `%s`.  %s.\n\n" % (md5, rec["correction"]))
    f.write("| kernel | read (MB) | written (MB) | dispatches |\n|---|---|---|---|\n")
    for k, v in kernels.items():
      part = "" if k in SEQUENCE else " (not part of the launch sequence)"
      f.write("| %s%s | %.1f | %.1f | %d |\n" % (k, part, v["read"] / 1e6, v["written"] / 1e6, v["dispatches"]))
    f.write("\nLaunch sequence (k_strip_scatter + k_strip_combine): **%.1f MB** against **%.1f MB** algorithmic "
            "(%.2fx).\n" % (seq / 1e6, alg / 1e6, seq / alg))
  print(json.dumps(rec, indent=1))


if __name__ == "__main__":
  main()
