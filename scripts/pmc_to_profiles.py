#!/usr/bin/env python3
"""gpurun_out/pmc_* (rocprofv3 --pmc passes over `bench.py --steps 3`) -> profiles/pmc_summary.json (per-launch HBM bytes).
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled
(MI355X_MICROARCH.md 'HBM').  Values are means over the dispatches of the bench run (poses differ slightly)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
summ = json.loads(subprocess.check_output([sys.executable, os.path.join(root, "scripts", "pmc_summarize.py"), os.path.join(root, "gpurun_out")]))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), python3 bench.py --steps 3 --warmup 1", "raw": summ}
for prefix, name in (("march_kernel<false", "march"), ("shade_kernel<0, 0", "shade"), ("composite_kernel", "composite")):
    for key in summ:
        if key.startswith(prefix) and "FETCH_SIZE" in summ[key] and "WRITE_SIZE" in summ[key]:
            out[f"{name}_hbm_bytes_per_launch"] = (2.0 * summ[key]["FETCH_SIZE"] + summ[key]["WRITE_SIZE"]) * 1024.0
json.dump(out, open(os.path.join(root, "profiles", "pmc_summary.json"), "w"), indent=1)
print({k: v for k, v in out.items() if k.endswith("per_launch")})
