#!/usr/bin/env python3
"""GPU timing of the candidates for the speculative schedule's guess stage on the roofline shape
(65,536 x 4096 complex64): round 2's windowed-L1 kernel against truncated coarse spectra (first M samples of every
row, zero-filled to 2M, maxima only -- the fused kernel with in_stride > n_in), and the cost of an exact check of K
rows (the same arg-max-only pass on the first K rows)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import _lib  # noqa: E402
from xmris_amd import device as dev  # noqa: E402

nv, nt, N = 65536, 4096, 8192
x = torch.randn((nv, nt, 2), device="cuda", dtype=torch.float32)
x = torch.view_as_complex(x).contiguous()
win = torch.from_numpy(np.exp(-np.pi * 5.0 * np.arange(N) / 5000.0)).to("cuda", torch.float32)
st = torch.cuda.current_stream().cuda_stream
FL = _lib.XM_FFT_ORTHO | _lib.XM_FFT_SHIFT_OUT | _lib.XM_AMAX_VALUE_ONLY


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # microseconds


key = dev.new_argmax_key(x.device)
norm = torch.empty(nv, dtype=torch.float32, device="cuda")
idx = torch.zeros(nv, dtype=torch.int32, device="cuda")
print(f"row_l1 (n_used 2304, sub_step 8, key):   {timeit(lambda: dev.row_l1(x, win, 0, n_used=2304, sub_step=8, key=key)):8.1f} us")
print(f"row_l1 (n_used 2304, sub_step 8, norms): {timeit(lambda: dev.row_l1(x, win, 0, out=norm, n_used=2304, sub_step=8)):8.1f} us")
est = torch.empty(nv, dtype=torch.float32, device="cuda")
wkey = dev.new_argmax_key(x.device)
hmax = torch.zeros(1, dtype=torch.float32, pin_memory=True)
hflat = torch.zeros(1, dtype=torch.int64, pin_memory=True)
row = torch.empty((1, nt), dtype=torch.complex128, device="cuda")
print(f"xm_guess_rows (coarse spectra, est + key):  {timeit(lambda: dev.guess_rows(x, N, win, est, key)):8.1f} us")
x128 = x[:32768].to(torch.complex128)
est2 = torch.empty(32768, dtype=torch.float32, device="cuda")
print(f"xm_guess_rows on complex128 rows (32768):    {timeit(lambda: dev.guess_rows(x128, N, win, est2, key)):8.1f} us")
del x128
key.zero_()


def both():
    dev.guess_rows(x, N, win, est, key)
    dev.guess_refine(x, N, win, est, key, wkey, hmax, hflat, row)


print(f"xm_guess_rows + xm_guess_refine (noise rows: every row is a candidate, 16 per workgroup): {timeit(both):8.1f} us")
for m in (256, 512, 1024, 2048):
    def coarse_rows(m=m):
        _lib.call("xm_pipeline_fused", x.data_ptr(), nt, None, win.data_ptr(), None, nv, m, 2 * m, 0, FL, norm.data_ptr(),
                  idx.data_ptr(), _lib.XM_C64, st)

    def coarse_key(m=m):
        _lib.call("xm_pipeline_fused", x.data_ptr(), nt, None, win.data_ptr(), None, nv, m, 2 * m, 0,
                  FL | _lib.XM_AMAX_GLOBAL_KEY, key.data_ptr(), None, _lib.XM_C64, st)

    try:
        print(f"coarse FFT first {m:4d} -> {2 * m:4d} bins, per-row maxima: {timeit(coarse_rows):8.1f} us")
    except Exception as e:  # noqa: BLE001
        print(f"coarse FFT {m}: per-row maxima unavailable ({e})")
    try:
        print(f"coarse FFT first {m:4d} -> {2 * m:4d} bins, key only:       {timeit(coarse_key):8.1f} us")
    except Exception as e:  # noqa: BLE001
        print(f"coarse FFT {m}: key unavailable ({e})")
for k in (64, 256, 512, 1024, 2048, 4096):
    def exact(k=k):
        _lib.call("xm_pipeline_fused", x.data_ptr(), nt, None, win.data_ptr(), None, k, nt, N, 0,
                  FL | _lib.XM_AMAX_GLOBAL_KEY, key.data_ptr(), None, _lib.XM_C64, st)

    print(f"exact arg-max-only pass on {k:5d} rows (key):  {timeit(exact):8.1f} us")
