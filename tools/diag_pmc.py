#!/usr/bin/env python3
"""Reduce tools/diag_pmc.sh's counter passes: per kernel name, the mean per dispatch of every counter (rocprofv3 counter_collection.csv)."""
import csv, glob, json, os, sys, collections
root = sys.argv[1]
out = {}
for p in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    name = os.path.relpath(p, root).split(os.sep)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    out[name] = {k: {"dispatches": len(cnt[k]), **{c: v / len(cnt[k]) for c, v in acc[k].items()}} for k in acc}
json.dump(out, sys.stdout, indent=1)
