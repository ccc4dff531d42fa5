#!/usr/bin/env python3
"""Offline study (numpy, CPU): how well do cheap per-row statistics rank the row that holds the global
max |X| of a dataset?  Backs the choice of the speculative schedule's guess stage.

For every dataset of the heterogeneous family (bench.py: synth_hetero -- per-voxel random line count 1-8,
widths 2-60 Hz, noise-only rows, one lipid-like row with the largest L1 norm) it prints the RANK of the true
arg-max row under each statistic (1 = the statistic's own winner, i.e. a guess that needs no candidates), and
how many rows lie within the statistic's error band of its maximum (= candidates an exact check must look at).

    python scripts/study_guess_statistics.py [--voxels 16384] [--seeds 6]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def hetero_numpy(nv, nt, dt, seed):
    """numpy twin of bench.synth_hetero (same recipe, host RNG): returns complex64 [nv, nt]."""
    rng = np.random.default_rng(seed)
    t = np.arange(nt) * dt
    x = np.zeros((nv, nt), dtype=np.complex64)
    nl = rng.integers(1, 9, size=nv)
    gain = rng.uniform(0.5, 1.5, size=nv)
    noise_only = rng.uniform(size=nv) < 0.05
    for s in range(0, nv, 1024):
        e = min(nv, s + 1024)
        acc = np.zeros((e - s, nt), dtype=np.complex128)
        for j in range(8):
            on = (nl[s:e] > j) & ~noise_only[s:e]
            a = rng.uniform(0.2, 1.0, size=e - s) * gain[s:e] * on
            w = rng.uniform(2.0, 60.0, size=e - s)
            f = rng.uniform(-2000.0, 2000.0, size=e - s)
            ph = rng.uniform(0, 2 * np.pi, size=e - s)
            acc += (a * np.exp(1j * ph))[:, None] * np.exp((-np.pi * w + 2j * np.pi * f)[:, None] * t[None, :])
        acc += (rng.standard_normal((e - s, nt)) + 1j * rng.standard_normal((e - s, nt))) * (0.02 / np.sqrt(2))
        x[s:e] = acc
    # lipid-like row: eight broad lines, 90 Hz apart -- the largest L1 norm of the dataset, not the tallest peak
    lip = int(rng.integers(0, nv))
    row = np.zeros(nt, dtype=np.complex128)
    for j in range(8):
        row += 3.0 * np.exp((-np.pi * 50.0 + 2j * np.pi * (-600.0 + 90.0 * j)) * t)
    x[lip] = row
    return x, t, lip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=16384)
    ap.add_argument("--seeds", type=int, default=6)
    args = ap.parse_args()
    nt, N, dt, lb = 4096, 8192, 1 / 5000.0, 5.0
    tt = np.arange(N) * dt
    win = np.exp(-np.pi * lb * tt)
    for seed in range(args.seeds):
        x, t, lip = hetero_numpy(args.voxels, nt, dt, 1000 + seed)
        nv = x.shape[0]
        true = np.empty(nv)
        for s in range(0, nv, 2048):
            xs = x[s:s + 2048].astype(np.complex128) * win[:nt]
            true[s:s + 2048] = np.abs(np.fft.fft(xs, n=N, axis=1)).max(axis=1) / np.sqrt(N)
        star = int(np.argmax(true))
        order = np.sort(true)[::-1]
        print(f"seed {seed}: nv={nv} true arg-max row {star} peak {true[star]:.4f}, 2nd {order[1]:.4f} "
              f"(gap {1 - order[1] / order[0]:.2e}), lipid row {lip} peak {true[lip]:.4f}")

        top = np.argsort(true)[::-1][:200]  # the statistic's accuracy where it matters: the 200 tallest rows

        def report(name, stat, band):
            rank = int((stat > stat[star]).sum()) + 1
            within = int((stat >= band * stat.max()).sum())
            ratio = stat[top] / true[top]
            need = ratio.min() / ratio.max()  # the band that is GUARANTEED to hold the true row on this dataset
            print(f"   {name:46s} rank of true row {rank:6d}   rows >= {band:.2f} max: {within:6d}   "
                  f"est/true top-200 [{ratio.min():.3f}, {ratio.max():.3f}] -> band {need:.3f}: "
                  f"{int((stat >= need * stat.max()).sum()):6d} rows")

        # (1) today's guess: windowed L1 over every 8th 128-sample block of the first 2304 samples
        wabs = win[:nt].astype(np.float32)
        blocks = [b for b in range(0, 2304 // 128) if b % 8 == 0]
        idx = np.concatenate([np.arange(b * 128, b * 128 + 128) for b in blocks])
        l1 = (np.abs(x[:, idx]) * wabs[idx]).sum(axis=1) / np.sqrt(N)
        report("L1 subsample (round 2)", l1, 0.5)
        l1f = (np.abs(x[:, :2304]) * wabs[:2304]).sum(axis=1) / np.sqrt(N)
        report("L1 over the window's support", l1f, 0.5)
        # (2) truncated coarse spectra: first M samples, G-point grid, fp32
        for M, G in ((256, 512), (256, 1024), (512, 1024), (512, 2048), (1024, 2048)):
            xs = (x[:, :M] * wabs[:M]).astype(np.complex64)
            p = np.abs(np.fft.fft(xs, n=G, axis=1)) ** 2
            e = np.sqrt(p.max(axis=1)) / np.sqrt(N)
            report(f"coarse FFT: first {M} samples, {G}-bin grid", e, 0.9)
            # (3) three-point refinement: 1/|X|^2 of a Lorentzian line is a parabola in frequency
            k = p.argmax(axis=1)
            r = np.arange(nv)
            ym, y0, yp = 1.0 / p[r, (k - 1) % G], 1.0 / p[r, k], 1.0 / p[r, (k + 1) % G]
            den = ym - 2 * y0 + yp
            delta = np.where(den > 0, 0.5 * (ym - yp) / np.where(den > 0, den, 1), 0.0)
            delta = np.clip(delta, -0.5, 0.5)
            ymin = y0 - 0.25 * (ym - yp) * delta
            e3 = np.sqrt(1.0 / np.maximum(ymin, 1e-30)) / np.sqrt(N)
            report("   ... + inverse-parabola peak", e3, 0.95)


if __name__ == "__main__":
    main()
