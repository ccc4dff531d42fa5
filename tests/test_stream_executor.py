"""CPU tests of the streaming EXECUTOR (`xmris_amd.pipeline.run_stream(speculate=True)`), the code that BASELINE
configs[3] (524,288 voxels over 8 GPUs) stands on: look-ahead, the order of the exchange / broadcast calls, verification,
cross-rank repairs, hedged searches, the hand-off to the device search.  The device entry points and the CUDA stream /
event objects are replaced by numpy stand-ins (`tests/_stream_double.py`); the ranks are spawned processes that talk
through gloo + `sharding.ShmExchange`, exactly as `bench.py --gpus 8` wires them.

Reference statement served: ONE global arg-max and ONE (p0, p1) per dataset (phasing.py:229, 276-290) -- every rank
must end with the single-process result of the non-speculative ("classic") schedule, bit for bit."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))

N_IN, N_OUT, LB, DT = 1024, 2048, 5.0, 1.0 / 5000.0
ROWS_PER_RANK, N_SETS = 8, 14
MISSES = {4: (2, 6), 9: (5, 3)}  # dataset -> (rank of the burst row = the true arg-max, rank of the coarse winner)
if os.environ.get("XM_TEST_MISSES"):  # "dataset:true_rank:coarse_rank,..." -- another placement (see the last test)
    MISSES = {int(a): (int(b), int(c)) for a, b, c in (tok.split(":") for tok in os.environ["XM_TEST_MISSES"].split(","))}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_dataset(d: int, world: int):
    """[world * ROWS_PER_RANK, N_IN] complex128 FIDs of dataset d.  The brightest decaying voxel rotates over the ranks
    1 .. world-1 (rank 0 never owns a winner); datasets in MISSES also hold a row whose signal starts after sample 600
    -- invisible to the coarse spectra (first 512 samples), the tallest line of the dataset -- on ANOTHER rank."""
    rows = world * ROWS_PER_RANK
    rng = np.random.default_rng(700 + d)
    t = np.arange(N_IN) * DT
    amp = 0.5 + rng.random(rows)
    f0 = rng.uniform(-1500, 1500, rows)
    x = amp[:, None] * np.exp(-25.0 * t)[None, :] * np.exp(2j * np.pi * f0[:, None] * t[None, :])
    x = x + 0.3 * amp[:, None] * np.exp(-40.0 * t)[None, :] * np.exp(2j * np.pi * (f0[:, None] + 400.0) * t[None, :] + 0.8j)
    x = x + 0.01 * (rng.standard_normal((rows, N_IN)) + 1j * rng.standard_normal((rows, N_IN)))
    if world > 1:
        star_rank = 1 + d % (world - 1)
        if d in MISSES:
            star_rank = MISSES[d][1] % world
    else:
        star_rank = 0
    star = star_rank * ROWS_PER_RANK + d % ROWS_PER_RANK
    x[star] *= 3.0 / amp[star]
    if d in MISSES:
        burst = (MISSES[d][0] % world) * ROWS_PER_RANK + (d + 3) % ROWS_PER_RANK
        late = np.zeros(N_IN)
        late[600:1000] = 1.0
        x[burst] = 40.0 * late * np.exp(2j * np.pi * 333.0 * t + 0.4j) + 0.01 * rng.standard_normal(N_IN)
    return x


def _plan(pl, torch):
    t = np.arange(N_IN) * DT
    return pl.make_plan(torch.zeros((1, N_IN), dtype=torch.complex128), t, N_OUT, LB)


def _digest(t):
    return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()


def reference_results(world: int):
    """The classic (non-speculative) schedule on the WHOLE datasets in this process."""
    import torch

    from xmris_amd import pipeline as pl

    plan = _plan(pl, torch)
    ins = [torch.from_numpy(make_dataset(d, world)) for d in range(N_SETS)]
    outs = [torch.empty((ins[0].shape[0], N_OUT), dtype=torch.complex128) for _ in range(N_SETS)]
    res = pl.run_stream(ins, outs, plan, speculate=False)
    return res, outs


def test_speculative_executor_equals_classic_single_process(monkeypatch):
    """One process, both search engines: hits, two repaired misses, results and outputs equal to the classic schedule
    bit for bit; the device engine searches every dataset behind the fill with `xm_search_launch`."""
    import _stream_double

    _stream_double.install(monkeypatch)
    import torch

    from xmris_amd import pipeline as pl

    monkeypatch.setenv("XM_SOLVER_THREADS", "2")
    ref, ref_out = reference_results(1)
    for engine in ("device", "host"):
        monkeypatch.setenv("XMRIS_AMD_SEARCH", engine)
        for k in _stream_double.COUNTS:
            _stream_double.COUNTS[k] = 0
        plan = _plan(pl, torch)
        ins = [torch.from_numpy(make_dataset(d, 1)) for d in range(N_SETS)]
        outs = [torch.empty((ins[0].shape[0], N_OUT), dtype=torch.complex128) for _ in range(N_SETS)]
        got = pl.run_stream(ins, outs, plan, speculate=True)
        for d in range(N_SETS):
            assert (got[d].p0, got[d].p1, got[d].pivot, got[d].flat_index) == (ref[d].p0, ref[d].p1, ref[d].pivot, ref[d].flat_index), (engine, d)
            assert got[d].speculation == ("repaired" if d in MISSES else "hit"), (engine, d, got[d].speculation)
            assert torch.equal(outs[d], ref_out[d]), (engine, d)
        launched = _stream_double.COUNTS["search_launch"]
        assert (launched >= N_SETS - 5) if engine == "device" else launched == 0, (engine, launched)
        # launches with an output: one main pass per dataset (+ the winner's fp64 spectrum with XMRIS_AMD_SLICE=device:
        # by default the host computes it with the reference's numpy statements), both again for a repair
        per_set = 1 if pl.slice_on_host() else 2
        assert _stream_double.COUNTS["guess_rows"] == N_SETS and _stream_double.COUNTS["main"] == per_set * (N_SETS + len(MISSES))


@pytest.mark.parametrize("cpus,ranks,gpus,want_device", [(16, 1, 1, False), (1, 1, 1, True), (16, 6, 1, False), (8, 6, 1, False),
                                                        (8, 8, 8, True), (16, 8, 8, False), (128, 8, 8, False)])
def test_auto_engine_rule(monkeypatch, cpus, ranks, gpus, want_device):
    """`XMRIS_AMD_SEARCH=auto`: the device-resident search only where a rank has a GPU of its own AND fewer than two
    CPUs -- the six-rank rehearsal on one card (16 CPUs) had picked it and ran 100 x slower (CU-masked queues of six
    processes on one GPU), profiles/r04/rehearsal_6ranks.txt."""
    import _stream_double

    _stream_double.install(monkeypatch)
    import torch

    from xmris_amd import autophase_solver as aps
    from xmris_amd import pipeline as pl

    monkeypatch.setenv("XM_SOLVER_THREADS", "2")
    monkeypatch.delenv("XMRIS_AMD_SEARCH", raising=False)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", str(ranks))
    monkeypatch.setattr(aps, "_CPU_SHARE", cpus)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: gpus)
    for k in _stream_double.COUNTS:
        _stream_double.COUNTS[k] = 0
    plan = _plan(pl, torch)
    ins = [torch.from_numpy(make_dataset(d, 1)) for d in range(N_SETS)]
    outs = [torch.empty((ins[0].shape[0], N_OUT), dtype=torch.complex128) for _ in range(N_SETS)]
    pl.run_stream(ins, outs, plan, speculate=True)  # (one process: `exchange` is None, LOCAL_WORLD_SIZE only feeds the rule)
    assert (_stream_double.COUNTS["search_launch"] > 0) == want_device


def _rank_main(rank, world, port, engine, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_WORLD_SIZE=str(world), XM_SOLVER_THREADS="1", XMRIS_AMD_SEARCH=engine,
                      XM_POLISH_THREADS="1")  # (polish helpers on a thread: eight ranks x four worker processes is a crowd)
    if engine == "host":
        os.environ["XM_HEDGE_SPACING"] = "1"  # (a rank owns every seventh dataset here: the one-in-eight rule would never let it hedge twice)
        os.environ["XM_TEST_SLOW_SEARCH"] = "7,4000"  # dataset 7's search reaches its owner's search service 4 s late: started a second time
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import _stream_double

    _stream_double.install()
    import torch
    import torch.distributed as dist

    from xmris_amd import pipeline as pl
    from xmris_amd import sharding

    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shm = sharding.ShmExchange.create(dist)

        def exchange(amax, gflat):
            owner, g, _ = shm.exchange_argmax(amax, gflat)
            return owner == rank, g, owner

        plan = _plan(pl, torch)
        lo, hi = sharding.shard_bounds(world * ROWS_PER_RANK, world, rank)
        ins = [torch.from_numpy(make_dataset(d, world)[lo:hi].copy()) for d in range(N_SETS)]
        outs = [torch.empty((hi - lo, N_OUT), dtype=torch.complex128) for _ in range(N_SETS)]
        trace = []
        res = pl.run_stream(ins, outs, plan, speculate=True, exchange=exchange, broadcast=shm.broadcast_params,
                            rank_offset_rows=lo, trace=trace)
        q.put((rank, [(r.p0, r.p1, r.pivot, r.flat_index, r.speculation, r.owner, r.mine, r.hedged) for r in res],
               [_digest(o) for o in outs], dict(shm.calls), dict(_stream_double.COUNTS)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("engine", ["device", "host"])
def test_world_8_speculative_executor(monkeypatch, engine):
    """EIGHT ranks (spawned processes, gloo + the shared-memory exchange), 14 datasets, `overlap` on: the winners rotate
    over ranks 1..7 (rank 0 never owns one), two datasets are guessed wrong with the guessed row and the true row on
    different ranks (cross-rank repair), and with the host engine one search reaches its service 4 s late and is started a second
    time.  Every rank must return the single-process classic result bit for bit -- (p0, p1), pivot, flat index, and
    its shard of every output -- and all ranks must have made the same number of exchange and broadcast calls."""
    import _stream_double
    import torch.multiprocessing as mp

    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, engine, q)) for r in range(world)]
    for p in procs:
        p.start()
    _stream_double.install(monkeypatch)
    monkeypatch.setenv("XM_SOLVER_THREADS", "1")
    ref, ref_out = reference_results(world)
    got = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    owners = [ref[d].flat_index // N_OUT // ROWS_PER_RANK for d in range(N_SETS)]
    assert 0 not in owners and len(set(owners)) >= 6, owners
    for d, (true_rank, _) in MISSES.items():
        assert owners[d] == true_rank
    calls = [g[3] for g in got]
    assert all(c == calls[0] for c in calls), calls  # the same sequence length on every rank
    assert calls[0]["gather"] >= 2 * N_SETS and calls[0]["broadcast"] >= N_SETS + len(MISSES)
    hedged = 0
    for rank, res, digests, _, counts in got:
        lo, hi = rank * ROWS_PER_RANK, (rank + 1) * ROWS_PER_RANK
        for d in range(N_SETS):
            p0, p1, pivot, flat, spec, owner, mine, hg = res[d]
            assert (p0, p1, pivot, flat) == (ref[d].p0, ref[d].p1, ref[d].pivot, ref[d].flat_index), (engine, rank, d)
            assert owner == owners[d] and mine == (owner == rank), (rank, d, owner)
            assert spec == ("repaired" if d in MISSES else "hit"), (rank, d, spec)
            assert digests[d] == hashlib.sha256(np.ascontiguousarray(ref_out[d][lo:hi].numpy()).tobytes()).hexdigest(), (rank, d)
            hedged += int(bool(hg))
        if engine == "device":  # behind the fill (the first datasets of a call), every search a rank owns is a search kernel
            # (of the GUESSED winner: a missed dataset's kernel ran on the coarse winner's rank, its repair on the host)
            owned = [d for d in range(N_SETS) if (MISSES[d][1] if d in MISSES else owners[d]) == rank]
            assert len([d for d in owned if d >= 5]) <= counts["search_launch"] <= len(owned), (rank, counts, owned)
        else:
            assert counts["search_launch"] == 0
    if engine == "host":
        assert hedged >= 1  # the late search of dataset 7


def test_world_8_with_the_misses_inside_the_pipeline_fill():
    """The same eight-rank run with the two wrongly guessed datasets at the very START of the call (datasets 0 and 1): the
    repairs then fall into the phase in which the look-ahead is still being built up (`fill_ramp`: three searches before
    the first main pass, three more with every dataset), where the order of exchange / broadcast calls differs from
    the steady state's.  Runs this module's world-8 test (host engine) in a child pytest with XM_TEST_MISSES set."""
    import subprocess

    env = dict(os.environ, XM_TEST_MISSES="0:2:6,1:5:3")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k", "test_world_8_speculative_executor and host",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
