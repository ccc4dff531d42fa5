#!/usr/bin/env python3
"""(A measurement tool that uses the CPU oracle as its checker: it lives under tests/, the only place besides
__graft_entry__.smoke() and bench.py's cpu_baseline leg that may import oracle/.)
How often are (p0, p1) of the HIP path EXACTLY the CPU oracle's?  Random datasets of several families (a few damped
lines, many lines, noise only, one voxel far brighter, short and long FIDs), both storage precisions, the fused
zero_fill -> apodize_exp -> to_spectrum -> autophase through `pipeline.run` against `oracle.pipeline_values` on the SAME
array.  Since round 4 the search runs on the reference's slice bit for bit (`pipeline.winner_spectrum`), its generations
replicate scipy's and its polish follows scipy's route; what could still differ is a generation's accept / reject
decision on a near-tie (the native objective differs from numpy's in the last bits).  Prints one line per case and a
summary.  usage: tool_sweep_autophase_exact.py [seed=0] [cases=40]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import xmris_oracle as orc  # noqa: E402  (the checker)
from xmris_amd import device as dev  # noqa: E402
from xmris_amd import pipeline as pipe  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
exact = total = 0
worst_dp = worst_err = {"complex64": 0.0, "complex128": 0.0}
worst_dp, worst_err = dict(worst_dp), dict(worst_err)
t_begin = time.time()
for case in range(n_cases):
    fam = rng.choice(["lines", "many", "noise", "bright", "short"])
    n_in = int(rng.choice([1024, 2048, 4096])) if fam != "short" else int(rng.choice([512, 600, 1000]))
    ratio = int(rng.choice([1, 2, 2, 4]))
    n_out = n_in * ratio if ratio > 1 else n_in
    if n_out not in (512, 600, 1000, 1024, 2048, 4096, 8192, 16384, 1200, 2000, 2400, 4000):
        n_out = 2048
    nv = int(rng.integers(3, 40))
    dt = 1.0 / float(rng.choice([2000.0, 5000.0, 10000.0]))
    t = np.arange(n_in) * dt
    lb = float(rng.choice([0.0, 2.0, 5.0, 12.0]))
    x = 0.02 * (rng.standard_normal((nv, n_in)) + 1j * rng.standard_normal((nv, n_in)))
    if fam != "noise":
        n_lines = {"lines": 3, "many": 12, "bright": 2, "short": 2}[fam]
        for _ in range(n_lines):
            a = rng.uniform(0.2, 1.0, nv)[:, None]
            x += a * np.exp((-np.pi * rng.uniform(2, 50) + 2j * np.pi * rng.uniform(-0.4, 0.4) / dt) * t)[None, :] * np.exp(1j * rng.uniform(-3, 3))
        if fam == "bright":
            x[rng.integers(0, nv)] *= 6.0
    else:
        x *= 50.0
    for dtype in ("complex128", "complex64"):
        xs = x.astype(dtype)
        ref, info = orc.pipeline_values(xs, t, n_out, lb, peak_width=100)
        out, res, _ = pipe.run(dev.to_device(xs), t, n_out, lb)
        dp = max(abs(res.p0 - info["p0"]), abs(res.p1 - info["p1"]))
        err = float(np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max())
        same = dp == 0.0 and res.flat_index == info["flat_idx"]
        exact += same
        total += 1
        worst_dp[dtype] = max(worst_dp[dtype], dp)
        worst_err[dtype] = max(worst_err[dtype], err)
        print(f"case {case:3d} {fam:6s} {dtype:10s} {nv:3d} x {n_in:5d} -> {n_out:5d} lb {lb:4.1f}: nfev {res.nfev:5d} / {info['nfev']:5d}  "
              f"|dp| {dp:.3e}  spectrum rel err {err:.3e}  {'EXACT' if same else 'differs'}")
if len(sys.argv) > 3 and sys.argv[3] == "methods":
    # the other objectives and the p0-only search (phasing.py:100-157, 257-274), on the oracle's own slice: the
    # generations of the two ROI scores sit on plateaus and kinks, where a last-bit difference between the native and the
    # numpy objective CAN flip an accept / reject decision -- how often does it?
    rng = np.random.default_rng(1000 + int(sys.argv[1]) if len(sys.argv) > 1 else 1000)
    m_exact = m_total = 0
    for case in range(max(4, n_cases // 3)):
        nv, n_in, n_out = int(rng.integers(3, 20)), 1024, 2048
        t = np.arange(n_in) * 2e-4
        x = 0.02 * (rng.standard_normal((nv, n_in)) + 1j * rng.standard_normal((nv, n_in)))
        for _ in range(3):
            x += rng.uniform(0.2, 1.0, nv)[:, None] * np.exp((-np.pi * rng.uniform(3, 40) + 2j * np.pi * rng.uniform(-2000, 2000)) * t)[None, :] * np.exp(1j * rng.uniform(-3, 3))
        _, info = orc.pipeline_values(x, t, n_out, 5.0, peak_width=100, solve=False)
        from xmris_amd.autophase_solver import index_width_of

        iw = index_width_of(info["freq"], 100)
        for method in ("acme", "peak_minima", "positivity"):
            for p0_only in (False, True):
                ref = orc.autophase_solve(info["slice"], info["freq"], info["pivot"], info["target_idx"], iw, method=method, p0_only=p0_only)
                out, res, _ = pipe.run(dev.to_device(x), t, n_out, 5.0, method=method, p0_only=p0_only)
                dp = max(abs(res.p0 - ref[0]), abs(res.p1 - ref[1]))
                m_exact += dp == 0.0
                m_total += 1
                print(f"methods case {case:2d} {method:11s} p0_only={int(p0_only)}: |dp| {dp:.3e}  {'EXACT' if dp == 0.0 else 'differs'}")
    print(f"methods: {m_exact} of {m_total} exactly the oracle's")
print(f"{exact} of {total} (p0, p1) pairs EXACTLY the oracle's; worst |dp| complex128 {worst_dp['complex128']:.3e} deg, complex64 "
      f"{worst_dp['complex64']:.3e} deg; worst spectrum rel err complex128 {worst_err['complex128']:.3e}, complex64 {worst_err['complex64']:.3e}"
      f"  ({time.time() - t_begin:.0f} s)")
