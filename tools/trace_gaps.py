"""Kernel durations and the gaps between consecutive kernels from a rocprofv3 --kernel-trace CSV (last N rows)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[len(rows) - n - skip: len(rows) - skip]
prev = None
for r in rows:
  s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
  print("%-60s dur %6.2f us  gap %6.2f us  grid %s wg %s" % (r["Kernel_Name"][:60], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
  prev = e
