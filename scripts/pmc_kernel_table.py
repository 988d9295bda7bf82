#!/usr/bin/env python3
"""Mean per dispatch of every counter of the kernels whose name contains <substr>, over the rocprofv3 --pmc output directories given.
usage: python scripts/pmc_kernel_table.py <substr> gpurun_out/pmc_a gpurun_out/pmc_b ..."""
import csv, glob, os, sys
sub = sys.argv[1]
for d in sys.argv[2:]:
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if sub in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    print(d, {k: round(sum(v) / len(v), 1) for k, v in sorted(acc.items())}, "dispatches", max((len(v) for v in acc.values()), default=0))
