#!/usr/bin/env python3
"""Prints a window of a rocprofv3 trace as a timeline: start offset (us), duration (us), queue/stream, kernel or copy.
   tools/timeline.py <dir with *_kernel_trace.csv [and *_memory_copy_trace.csv]> [first_row_fraction=0.6] [rows=60]"""
import csv, glob, os, sys
src = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        if "Kernel_Name" in r:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"q{r.get('Queue_Id', '?'):>3} s{r.get('Stream_Id', '?'):>3}", r["Kernel_Name"][:70]))
        elif "Direction" in r:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), f"copy s{r.get('Stream_Id', '?'):>3}", r["Direction"]))
rows.sort()
i0 = int(len(rows) * frac)
t0 = rows[i0][0]
for s, e, where, what in rows[i0:i0 + count]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  {where}  {what}")
