#!/usr/bin/env python3
"""(A measurement tool that uses the CPU oracle as its checker: it lives under tests/, the only place besides
__graft_entry__.smoke() and bench.py's cpu_baseline leg that may import oracle/.)
BASELINE configs[0] (README quick start: 5 x 1024 noise FIDs, zero_fill(2048), apodize_exp(lb=5), to_spectrum,
autophase) end to end on the GPU against the CPU oracle: |dp0|, |dp1| and the relative error of the phased spectra for
the injected-parameter route and the own solve with each polish mode ("exact" is the default everywhere since round 4),
through the one-dataset call and through the streaming executor, both storage precisions.  Output kept under profiles/ (north_star: <= 1e-5)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import xmris_oracle as orc  # noqa: E402
from xmris_amd import device as dev  # noqa: E402
from xmris_amd import pipeline as pipe  # noqa: E402


def relerr(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


rng = np.random.default_rng(42)
t = np.linspace(0, 1, 1024)
x = rng.standard_normal((5, 1024)) + 1j * rng.standard_normal((5, 1024))
ref, info = orc.pipeline_values(x.astype(np.complex128), t, 2048, 5.0, peak_width=100)
print(f"oracle: flat index {info['flat_idx']}, (p0, p1) = ({info['p0']:.9f}, {info['p1']:.9f}), nfev {info['nfev']}")
ref_c128, info_c128 = ref, info
for dtype in ("complex128", "complex64", "complex64 vs the oracle on the SAME complex64 array"):
    if dtype.startswith("complex64 vs"):
        # what the reference computes when it is handed the complex64 array (numpy promotes it at the apodisation,
        # fid.py:136-139): the comparison that isolates the implementation from the rounding of the INPUT
        ref, info = orc.pipeline_values(x.astype(np.complex64), t, 2048, 5.0, peak_width=100)
        print(f"oracle on the complex64 array: (p0, p1) = ({info['p0']:.9f}, {info['p1']:.9f}), nfev {info['nfev']}")
        dtype = "complex64"
    else:
        ref, info = ref_c128, info_c128
    xd = dev.to_device(x.astype(dtype))
    out, res, _ = pipe.run(xd, t, 2048, 5.0, params=(info["p0"], info["p1"]))
    print(f"[{dtype}] oracle's (p0, p1) injected:          spectrum rel err {relerr(out.cpu().numpy(), ref):.3e}  "
          f"(flat index {'equal' if res.flat_index == info['flat_idx'] else 'DIFFERENT'})")
    for polish in ("native", "numpy", "exact"):
        out, res, _ = pipe.run(xd, t, 2048, 5.0, polish=polish)
        print(f"[{dtype}] pipeline.run,        polish={polish:6s}: |dp0| {abs(res.p0 - info['p0']):.3e} deg  |dp1| {abs(res.p1 - info['p1']):.3e} deg  "
              f"nfev {res.nfev}  spectrum rel err {relerr(out.cpu().numpy(), ref):.3e}  |X| rel err "
              f"{relerr(np.abs(out.cpu().numpy()), np.abs(ref)):.3e}")
    # the STREAMING executor (what bench.py times), default polish ("exact"), six copies of the dataset in a row
    import torch

    plan = pipe.make_plan(xd, t, 2048, 5.0)
    outs = [torch.empty((5, 2048), dtype=xd.dtype, device=xd.device) for _ in range(6)]
    results = pipe.run_stream([xd] * 6, outs, plan, speculate=True)
    worst = max(relerr(o.cpu().numpy(), ref) for o in outs)
    r = results[-1]
    print(f"[{dtype}] run_stream (6 datasets), polish=exact : |dp0| {abs(r.p0 - info['p0']):.3e} deg  |dp1| {abs(r.p1 - info['p1']):.3e} deg  "
          f"nfev {r.nfev}  spectrum rel err (worst of 6) {worst:.3e}")
