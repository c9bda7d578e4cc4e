#!/usr/bin/env python3
"""Print the last kernels of a rocprofv3 kernel trace as a timeline (us relative to the first shown)."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f -> %9.1f  (%7.1f us)  q%-3s %s" % (a, b, b - a, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
