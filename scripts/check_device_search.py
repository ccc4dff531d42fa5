"""Device-resident search (csrc/xm_search.hip) against the host engines on one GPU: the objective against the numpy
statement, the generations against xm_solver_de (bit for bit) and, on the golden slice, against scipy itself; timing."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from xmris_amd import autophase_solver as aps  # noqa: E402
from xmris_amd import device as dev  # noqa: E402


def make_slice(n, seed, sw=5000.0):
    rng = np.random.default_rng(seed)
    nt = n // 2
    t = np.arange(nt) / sw
    fid = np.zeros(nt, complex)
    for _ in range(int(rng.integers(3, 6))):
        fid += rng.uniform(0.3, 1.0) * np.exp(-rng.uniform(15.0, 60.0) * t) * np.exp(2j * np.pi * rng.uniform(-2000, 2000) * t)
    fid += 0.01 * (rng.standard_normal(nt) + 1j * rng.standard_normal(nt))
    freq = np.roll(np.fft.fftfreq(n, d=1 / sw), n // 2)
    spec = np.roll(np.fft.fft(np.pad(fid * np.exp(-np.pi * 3.0 * t), (0, n - nt)), norm="ortho"), n // 2)
    k = int(np.argmax(np.abs(spec)))
    spec = spec * np.exp(1j * aps.phase_angles(freq, rng.uniform(-150, 150), rng.uniform(-600, 600), float(freq[k])))
    return spec, freq, k


def run_search(sl_pinned, axis, rec, seq, p0_only, stream=None):
    dev.search_launch(sl_pinned, axis, rec, seq, p0_only=p0_only, stream=stream)
    t0 = time.perf_counter()
    while not dev.search_done(rec, seq):
        if time.perf_counter() - t0 > 20:
            raise TimeoutError("search did not finish")
    return dev.read_search_record(rec)


def main():
    torch.cuda.init()
    bad = 0
    rec = dev.new_search_record()
    seq = 0
    for n in (8192, 4096, 2048, 1536, 1000, 16384):
        for seed in range(3 if n != 8192 else 8):
            spec, freq, k = make_slice(n, 100 * n + seed)
            axis = dev.uniform_axis(freq)
            pinned = torch.from_numpy(spec.copy()).pin_memory()
            # objective
            rng = np.random.default_rng(seed)
            xs = np.stack([rng.uniform(-180, 180, 16), rng.uniform(-4000, 4000, 16)], 1)
            got = dev.search_eval(pinned, axis, xs)
            ref = np.array([aps.acme_score(x, spec, freq, float(freq[k])) for x in xs])
            err = np.abs(got - ref).max() / np.abs(ref).max()
            for p0_only in (False, True):
                seq += 1
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = run_search(pinned, axis, rec, seq, p0_only)
                dt = time.perf_counter() - t0
                obj = aps.NativeObjective(spec, freq, float(freq[k]), k, 1, "acme")
                rc, x, fun, nfev, nit = obj.de(p0_only)
                xd = np.array(r["x"][:1 if p0_only else 2])
                same = np.array_equal(xd, x) and r["nfev"] == nfev and r["nit"] == nit and r["target_idx"] == k and r["status"] == rc
                lo, hi = np.array([-180.0, -4000.0])[:len(x)], np.array([180.0, 4000.0])[:len(x)]
                _, g0 = obj.fg(np.clip(x, lo, hi), lo, hi)
                pg = np.where(g0 < 0, np.maximum(x - hi, g0), np.minimum(x - lo, g0))
                pgn = float(np.abs(pg).max())
                flag_same = r["needs_polish"] == (pgn > 0.5e-5)
                bad += int(not same) + int(not flag_same) + int(err > 1e-11)
                print(f"n={n:6d} seed={seed} p0_only={int(p0_only)} objective rel err {err:.1e}  search {1e3 * dt:6.2f} ms  nfev {r['nfev']:4d}/{nfev:4d} "
                      f"nit {r['nit']:2d}/{nit:2d}  x equal: {same}  dfun {abs(r['fun'] - fun) / abs(fun):.1e}  pg {r['pg_norm']:.2e}/{pgn:.2e} flag equal: {flag_same}  us/trial [point tables draw wait score]: "
                      + " ".join(f"{v / max(r['nfev'], 1):.2f}" for v in r["t_us"][:5]) + f"  total {r['t_us'][5] / 1e3:.2f} ms",
                      flush=True)
    # several searches at once on streams of their own: wall time per search
    spec, freq, k = make_slice(8192, 7)
    axis = dev.uniform_axis(freq)
    pinned = torch.from_numpy(spec.copy()).pin_memory()
    for conc in (1, 2, 4, 8):
        streams = [torch.cuda.Stream() for _ in range(conc)]
        recs = [dev.new_search_record() for _ in range(conc)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(conc):
            dev.search_launch(pinned, axis, recs[i], 1000 + conc, stream=streams[i])
        while not all(dev.search_done(recs[i], 1000 + conc) for i in range(conc)):
            pass
        print(f"{conc} searches at once: {1e3 * (time.perf_counter() - t0):.2f} ms wall", flush=True)
    print("MISMATCHES:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
