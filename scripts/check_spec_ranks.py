"""N-rank check of run_stream(speculate=True) on ONE shared GPU (run under torch.distributed.run; gloo + the shared-
memory exchange, exactly as bench.py wires its ranks): twelve datasets whose winners rotate over ranks 1 .. N-1 (rank 0
never owns one), two of them guessed wrong with the guessed row and the true row on different ranks.  Every rank's
(p0, p1), pivot and flat index and the gathered spectra must equal the ONE-rank classic schedule; all ranks must make
the same number of exchange / broadcast calls; per rank, the time from a dataset's exchange to the use of its (p0, p1)
is printed next to the look-ahead it has.   XMRIS_AMD_SEARCH=host|device selects the engine."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import device as dev, pipeline as pipe, sharding  # noqa: E402

os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
per, nt, target, n_sets = 64, 1024, 2048, 12
nv = per * world
t = np.arange(nt) * 2e-4
misses = {3: (world - 1, 1), 8: (1, world - 2)}  # dataset -> (rank of the burst row = true arg-max, rank of the coarse winner)
sets = []
for k in range(n_sets):
    rng = np.random.default_rng(50 + k)
    amp = 0.5 + rng.random(nv)
    f0 = rng.uniform(-1500, 1500, nv)
    x = amp[:, None] * np.exp(-25.0 * t)[None, :] * np.exp(2j * np.pi * f0[:, None] * t[None, :])
    x = x + 0.01 * (rng.standard_normal((nv, nt)) + 1j * rng.standard_normal((nv, nt)))
    star_rank = misses[k][1] if k in misses else 1 + k % (world - 1)
    star = star_rank * per + (7 * k) % per
    x[star] *= 3.0 / amp[star]
    if k in misses:
        burst = misses[k][0] * per + (11 * k) % per
        x[burst] = 0.01 * rng.standard_normal(nt)
        x[burst, 600:1000] = 60.0 * np.exp(2j * np.pi * 333.0 * t[600:1000])
    sets.append(x.astype(np.complex64))
lo, hi = sharding.shard_bounds(nv, world, rank)
mine = [dev.to_device(x[lo:hi]) for x in sets]
plan = pipe.make_plan(mine[0], t, target, 5.0)
outs = [torch.empty((hi - lo, target), dtype=torch.complex64, device="cuda") for _ in sets]
shm = sharding.ShmExchange.create(dist)


def exchange(amax, gflat):
    owner, gwin, _ = shm.exchange_argmax(amax, gflat)
    return owner == rank, gwin, owner


pipe.run_stream(mine[:4], outs[:4], plan, exchange=exchange, broadcast=shm.broadcast_params, rank_offset_rows=lo, speculate=True)
torch.cuda.synchronize()
dist.barrier()
trace = []
t0 = time.perf_counter()
res = pipe.run_stream(mine, outs, plan, exchange=exchange, broadcast=shm.broadcast_params, rank_offset_rows=lo, speculate=True,
                      trace=trace)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
lat = [1e3 * (e["t_solved"] - e["t_exchanged"]) for e in trace]
owned = sum(r.mine for r in res)
calls = [None] * world
dist.all_gather_object(calls, dict(shm.calls))
assert all(c == calls[0] for c in calls), calls
print(f"[rank {rank}] {1e3 * wall / n_sets:.3f} ms per dataset, searches owned {owned}, exchange-to-use latency mean {np.mean(lat):.2f} "
      f"max {np.max(lat):.2f} ms, exchange calls {calls[0]}", flush=True)
gathered = []
for o in outs:
    parts = [torch.empty((per, target), dtype=torch.complex64) for _ in range(world)]
    dist.all_gather(parts, o.cpu())
    gathered.append(torch.cat(parts).numpy())
summary = [None] * world
dist.all_gather_object(summary, [(r.p0, r.p1, r.pivot, r.flat_index, r.speculation, r.owner) for r in res])
if rank == 0:
    full = [dev.to_device(x) for x in sets]
    plan1 = pipe.make_plan(full[0], t, target, 5.0)
    refs = [torch.empty((nv, target), dtype=torch.complex64, device="cuda") for _ in sets]
    ref = pipe.run_stream(full, refs, plan1)  # one rank, the classic (non-speculative) schedule
    torch.cuda.synchronize()
    owners = []
    for k, b in enumerate(ref):
        want = (b.p0, b.p1, b.pivot, b.flat_index)
        for r in range(world):
            assert summary[r][k][:4] == want, (k, r, summary[r][k], want)
            assert summary[r][k][4] == ("repaired" if k in misses else "hit"), (k, r, summary[r][k])
        err = float(np.abs(gathered[k] - refs[k].cpu().numpy()).max() / np.abs(refs[k].cpu().numpy()).max())
        assert err < 1e-6, (k, err)
        owners.append(summary[0][k][5])
    assert 0 not in owners and len(set(owners)) == world - 1, owners
    print(f"{world}-rank speculative run_stream ({os.environ.get('XMRIS_AMD_SEARCH', 'auto')} search): OK, owners {owners}")
dist.destroy_process_group()
