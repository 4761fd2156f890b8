#!/usr/bin/env python3
"""Prints a window of a rocprofv3 --kernel-trace CSV as a timeline: start offset (us), duration (us), queue, kernel.
   tools/timeline.py <kernel_trace.csv> [first_row_fraction=0.6] [rows=60]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
i0 = int(len(rows) * frac)
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3} s{r.get('Stream_Id', '?'):>3}  {r['Kernel_Name'][:60]}")
