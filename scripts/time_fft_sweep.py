"""xm_fft1d_batched (ortho + fftshift, the to_spectrum seam) over the supported lengths, c64 and c128:
ms per call and GB/s of compulsory traffic (read + write once) on ~2 GiB (c64) / ~2 GiB (c128) of rows."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import device as dev
lengths = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 1536, 2048, 3072, 4096, 5120, 6144, 8192, 16384, 768, 1531, 1972, 2000, 4093, 1000, 6000]
for dt, B in ((torch.complex64, 8), (torch.complex128, 16)):
    for n in lengths:
        if not dev.fft_supported(n, dt == torch.complex128):
            print(f"{str(dt):18s} n={n:6d} unsupported"); continue
        nb = max(1, (1 << 30) // (n * B))
        x = torch.view_as_complex(torch.randn(nb, n, 2, device="cuda", dtype=torch.float32 if B == 8 else torch.float64))
        f = lambda: dev.fft(x, -1, shift_out=True)
        for _ in range(2): y = f()
        torch.cuda.synchronize(); ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            y = f(); e0.record()
            for _ in range(4): y = f()  # (back to back: the device's time per call, not the wrapper's)
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
        ms = float(np.median(ts))
        print(f"{str(dt):18s} n={n:6d} batch={nb:7d}  {ms:8.4f} ms  {2*nb*n*B/ms/1e6:8.1f} GB/s")
        del x, y
