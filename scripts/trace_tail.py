"""Print the last N proqa kernel launches of a rocprofv3 kernel-trace CSV (dev helper)."""
import csv
import sys

path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 19
rows = [r for r in csv.DictReader(open(path)) if "proqa" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-n]["Start_Timestamp"])
for r in rows[-n:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(r["Kernel_Name"][35:58].ljust(24), "%9.1f us  start %9.1f  end %9.1f  grid %s" % ((e - s) / 1e3, s / 1e3, e / 1e3, r["Grid_Size_X"]))
