#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (gpurun_out/pmc_<tag>/**/*counter_collection.csv) per kernel: mean per dispatch."""
import csv, glob, os, sys, json, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        out[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, cs in out.items():
    if not any(s in k for s in ("march", "shade", "composite", "ngp_")):
        continue
    res[k] = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    res[k]["_dispatches"] = max(len(v) for v in cs.values())
print(json.dumps(res, indent=1))
