#!/usr/bin/env python3
"""Host cost of the library's launches (the streaming executor's launch thread pays them once per dataset): each entry
point called in a loop on a tiny batch, wall time per call without waiting for the device."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import _lib  # noqa: E402
from xmris_amd import device as dev  # noqa: E402
from xmris_amd import pipeline as pl  # noqa: E402

nv = 64
for nt, N in ((4096, 8192), (1536, 1536), (2048, 4096)):
    x = torch.view_as_complex(torch.randn((nv, nt, 2), device="cuda", dtype=torch.float32)).contiguous()
    t = np.arange(nt) * 2e-4
    plan = pl.make_plan(x, t, N, 5.0)
    out = torch.empty((nv, N), dtype=x.dtype, device="cuda")
    win32 = plan.window.to(torch.float32) if plan.window.dtype != torch.float32 else plan.window
    est = torch.empty(nv, dtype=torch.float32, device="cuda")
    key, wkey = dev.new_argmax_key(x.device), dev.new_argmax_key(x.device)
    hmax = torch.zeros(1, dtype=torch.float32, pin_memory=True)
    hflat = torch.zeros(1, dtype=torch.int64, pin_memory=True)
    row = torch.empty((1, nt), dtype=torch.complex128, device="cuda")
    hs = torch.empty((1, N), dtype=torch.complex128, pin_memory=True)
    w64 = torch.from_numpy(np.ascontiguousarray(plan.window_host)).to("cuda", torch.float64)

    def timeit(name, fn, reps=300):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"{nt:5d} -> {N:5d}  {name:34s} {(t1 - t0) / reps * 1e6:7.1f} us per call")

    guess_ok = dev.guess_supported(x, N, plan.pad_left)
    if guess_ok:
        timeit("guess_rows", lambda: dev.guess_rows(x, N, win32, est, key))
        timeit("guess_refine", lambda: dev.guess_refine(x, N, win32, est, key, wkey, hmax, hflat, row))
    timeit("winner spectrum (1 row, c128)", lambda: dev.pipeline_fused(row, N, plan.pad_left, window=w64, out=hs))
    timeit("main pass (ramp)", lambda: pl.main_pass(plan, x, out, 10.0, -20.0, 0.0))
    e = torch.cuda.Event()
    timeit("torch Event.record", lambda: e.record())
    timeit("torch Event() + record", lambda: torch.cuda.Event().record())
    timeit("current_stream().cuda_stream", lambda: torch.cuda.current_stream(x.device).cuda_stream)
