#!/usr/bin/env python3
"""Turn the text output of scripts/pmc.sh (scripts/pmc_summary.py: mean counter values per kernel) into the small JSON
bench.py reads for `roofline.traffic`:
    python scripts/pmc_json.py profiles/r03/pmc_main_kernel.txt "k_zf2p<FftPlan<4096, 256, 16, 16, 16>, 13, 11>" 65536 4096 8192 > profiles/r03/pmc_main_kernel.json
The record names the kernel (spaces removed), the workload it ran on, FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports
them (bench.py applies the gfx950 x2 to FETCH_SIZE) and the commit the counters were collected at."""
import json
import subprocess
import sys

path, kernel, nv, nt, n_out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
want = kernel.replace(" ", "")
vals, cur = {}, None
for line in open(path):
    if not line.startswith(" "):
        cur = line.strip().replace("void ", "").replace(" ", "")
        continue
    if cur == want:
        name, value = line.split()[:2]
        vals[name] = float(value)
if "FETCH_SIZE" not in vals or "WRITE_SIZE" not in vals:
    sys.exit(f"{path}: no FETCH_SIZE / WRITE_SIZE for {kernel}")
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    commit = ""
print(json.dumps({"kernel": want, "voxels": nv, "n_time": nt, "target_points": n_out, "FETCH_SIZE_KB": vals["FETCH_SIZE"],
                  "WRITE_SIZE_KB": vals["WRITE_SIZE"], "source": path, "commit": commit or "unknown",
                  "counters": {k: v for k, v in sorted(vals.items())}}, indent=1))
