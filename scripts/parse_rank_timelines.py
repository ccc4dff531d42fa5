#!/usr/bin/env python3
"""stdin: stderr of `XM_BENCH_DEBUG=1 bench.py --gpus N ...` (the `[rank r] timeline ...` lines) -> per dataset, which
rank waited where for more than THRESH ms: selection event (sel), the other ranks' arrival at the exchange (exch), the
owner's result (solve)."""
import re
import sys

THRESH = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
ranks = {}
for line in sys.stdin:
    m = re.search(r"\[rank (\d+)\] timeline \([^)]*\) (.*)", line)
    if not m:
        continue
    sets = []
    for part in m.group(2).split(" |"):
        vals = re.findall(r"-?\d+\.\d|(?<![\d.])-(?![\d.])", part)[:5]
        if len(vals) == 5:
            sets.append([float(v) if v != "-" else float("nan") for v in vals])
    ranks[int(m.group(1))] = sets
n = min(len(v) for v in ranks.values())
for j in range(n):
    notes = []
    for r in sorted(ranks):
        st, sel, ex, col, sol = ranks[r][j]
        for name, d in (("sel", sel - st), ("exch", ex - sel), ("solve", sol - col)):
            if d > THRESH:
                notes.append(f"r{r} {name} {d:.1f}")
    t = ranks[min(ranks)][j]
    print(f"set {j:2d}  start {t[0]:8.1f}  solved {t[4]:8.1f}  " + ("; ".join(notes) if notes else "-"))
