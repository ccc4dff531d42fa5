"""`xm_zf_apod` (zero fill + apodisation in one launch) against the two staged launches, 65,536 x 4096 -> 8192."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xmris_amd import device as dev  # noqa: E402

nv, nt, N = 65536, 4096, 8192
w = np.exp(-np.pi * 5.0 * np.arange(N) / 5000.0)


def timed(fn, reps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, dt, promote in (("complex64 -> complex64", torch.complex64, False), ("complex64 -> complex128 (numpy's promotion)", torch.complex64, True),
                          ("complex128 -> complex128", torch.complex128, False)):
    rows = nv if not (promote or dt == torch.complex128) else nv // 2
    x = torch.randn((rows, nt), dtype=dt, device="cuda")
    esz_in = 8 if dt == torch.complex64 else 16
    esz_out = 16 if (promote or dt == torch.complex128) else 8
    alg = rows * (nt * esz_in + N * esz_out)
    ms = timed(lambda: dev.zf_apod(x, N, 0, w, promote=promote))
    if promote:
        ms2 = timed(lambda: dev.apodize(dev.zero_fill(x, 1, N).to(torch.complex128), 1, w))
    else:
        ms2 = timed(lambda: dev.apodize(dev.zero_fill(x, 1, N), 1, w))
    print(f"{name:45s} {rows} rows: one launch {ms:.3f} ms = {alg / ms / 1e9:.2f} TB/s of algorithmic traffic ({dev.last_kernel() or 'k_zf_apod'}); "
          f"staged zero_fill + apodize {ms2:.3f} ms ({ms2 / ms:.2f} x)")
    del x
