"""world_size-2 (and 4) CPU test (gloo) of the multi-GPU path: voxel-axis sharding with the O(1) arg-max
exchange and (p0, p1) broadcast of ``xmris_amd.sharding`` -- the same calls bench.py makes over RCCL.
Per-rank spectra come from numpy here (the kernels need a GPU); the sharding logic is what is tested:
the sharded result must equal the oracle run on the whole dataset, including an exact tie across ranks."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dataset(tie: bool):
    nv, nt = 12, 256
    t = np.arange(nt) / 2000.0
    rng = np.random.default_rng(5)
    amp = np.linspace(0.5, 1.5, nv)
    amp[8] = 3.0  # the winner lives in rank 1's shard
    x = amp[:, None] * (np.exp(-30 * t) * np.exp(2j * np.pi * 220 * t))[None, :]
    x = x + 0.01 * (rng.standard_normal((nv, nt)) + 1j * rng.standard_normal((nv, nt)))
    if tie:
        x[2] = x[8]  # identical spectrum in rank 0's shard: np.argmax must pick the FIRST (row 2)
    return x, t


def _worker(rank, world, port, tie, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from xmris_amd import autophase_solver as aps
    from xmris_amd import sharding

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x, t = _dataset(tie)
        n = 512
        lo, hi = sharding.shard_bounds(x.shape[0], world, rank)
        tt = t[0] + np.arange(n) * (t[1] - t[0])
        freq = np.roll(np.fft.fftfreq(n, d=tt[1] - tt[0]), n // 2)
        spec = np.roll(np.fft.fft(np.pad(x[lo:hi], ((0, 0), (0, n - x.shape[1]))) * np.exp(-np.pi * 5.0 * tt),
                                  axis=1, norm="ortho"), n // 2, axis=1)
        flat = int(np.argmax(np.abs(spec)))
        amax = float(np.abs(spec).reshape(-1)[flat])
        owner, gflat, gmax = sharding.exchange_argmax(amax, lo * n + flat, dist)
        # the single-node shared-memory exchange must agree with the collective one, dataset after dataset
        shm = sharding.ShmExchange.create(dist)
        assert shm.exchange_argmax(amax, lo * n + flat) == (owner, gflat, gmax)
        assert shm.broadcast_params([1.5 + owner, -2.0], owner) == [1.5 + owner, -2.0]
        rng = np.random.default_rng(100 + rank)
        for it in range(300):  # ranks run ahead of each other at random: two banks must be enough
            mine = (float(rng.integers(0, 4)), int(rng.integers(0, 1000)))
            ref = sharding.exchange_argmax(mine[0], mine[1], dist)
            if rng.random() < 0.3:
                import time
                time.sleep(float(rng.random()) * 1e-3)
            assert shm.exchange_argmax(*mine) == ref, it
            vals = shm.broadcast_params([it + 0.25, float(rank)], ref[0])
            assert vals == [it + 0.25, float(ref[0])], it
        k = gflat % n
        pivot = float(freq[k])
        params = [0.0, 0.0]
        if owner == rank:
            row = gflat // n - lo
            p0, p1, _ = aps.solve(spec[row], freq, pivot, k, aps.index_width_of(freq, 100))
            params = [p0, p1]
        p0, p1 = sharding.broadcast_params(params, owner, dist)
        out = spec * aps.phase_table(freq, p0, p1, pivot)[None, :]
        q.put((rank, lo, hi, owner, gflat, p0, p1, pivot, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("tie,world", [(False, 2), (True, 2), (True, 4)])
def test_two_rank_sharded_pipeline_matches_oracle(oracle, tie, world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, tie, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x, t = _dataset(tie)
    ref, info = oracle.pipeline_values(x, t, 512, 5.0, peak_width=100)
    per = 12 // world
    assert [(g[1], g[2]) for g in got] == [(r * per, (r + 1) * per) for r in range(world)]
    for g in got:
        assert g[4] == info["flat_idx"]
        assert g[3] == (0 if tie else 8 // per)  # the tie's first occurrence (row 2) lives on rank 0
        assert g[7] == info["pivot"]
        assert abs(g[5] - info["p0"]) < 1e-6 and abs(g[6] - info["p1"]) < 1e-6
    out = np.concatenate([g[8] for g in got], axis=0)
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
