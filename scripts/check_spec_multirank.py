"""2-rank check of run_stream(speculate=True) with a WRONG guess whose repair crosses ranks (run under
torch.distributed.run, ranks may share one GPU):  python -m torch.distributed.run --nproc-per-node 2 scripts/check_spec_multirank.py"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xmris_amd import device as dev, pipeline as pipe, sharding
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
nv, nt, target = 64, 1024, 2048
t = np.arange(nt) * 2e-4
rng = np.random.default_rng(5)
sets = []
for k in range(4):
    x = 0.01 * (rng.standard_normal((nv, nt)) + 1j * rng.standard_normal((nv, nt)))
    x[5] = 1.5 * np.exp(-20.0 * t) * np.exp(2j * np.pi * 310.0 * t)                   # rank 0: the guess stage's winner
    x[50] = 1.0 * np.exp(-20.0 * t) * np.exp(2j * np.pi * -120.0 * t)                 # rank 1: its guess stage's winner
    # (without it every row of rank 1 looks like noise to the coarse spectra, all of them become candidates and the
    # exact check finds the burst)
    # rank 1: the true maximum -- a burst in samples 600...999, which no guess stage reads (the coarse spectra look at
    # samples 0...511, the L1 subset at block 0 of every eight 128-sample blocks); the last dataset's burst is weak
    # and the guess is right
    x[40 + k, 600:1000] = (60.0 if k < 3 else 2.0) * np.exp(2j * np.pi * (-400.0 + 50 * k) * t[600:1000])
    sets.append(x.astype(np.complex64))
lo, hi = sharding.shard_bounds(nv, world, rank)
mine = [dev.to_device(x[lo:hi]) for x in sets]
plan = pipe.make_plan(mine[0], t, target, 5.0)
outs = [torch.empty((hi - lo, target), dtype=torch.complex64, device="cuda") for _ in sets]
shm = sharding.ShmExchange.create(dist)
def exchange(amax, gflat):
    owner, gwin, _ = shm.exchange_argmax(amax, gflat)
    return owner == rank, gwin, owner
res = pipe.run_stream(mine, outs, plan, exchange=exchange, broadcast=shm.broadcast_params, rank_offset_rows=lo, speculate=True)
torch.cuda.synchronize()
gathered = []
for o in outs:
    parts = [torch.empty((sharding.shard_bounds(nv, world, r)[1] - sharding.shard_bounds(nv, world, r)[0], target), dtype=torch.complex64) for r in range(world)]
    dist.all_gather(parts, o.cpu())
    gathered.append(torch.cat(parts).numpy())
if rank == 0:
    full = [dev.to_device(x) for x in sets]
    plan1 = pipe.make_plan(full[0], t, target, 5.0)
    refs = [torch.empty((nv, target), dtype=torch.complex64, device="cuda") for _ in sets]
    ref = pipe.run_stream(full, refs, plan1)
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(res, ref)):
        err = float(np.abs(gathered[k] - refs[k].cpu().numpy()).max() / np.abs(refs[k].cpu().numpy()).max())
        print(f"dataset {k}: {a.speculation:8s} winner row {a.flat_index // target} (ref {b.flat_index // target})  "
              f"p0 {a.p0:.9f} (ref {b.p0:.9f})  p1 {a.p1:.9f} (ref {b.p1:.9f})  out err {err:.2e}")
        assert a.flat_index == b.flat_index and a.p0 == b.p0 and a.p1 == b.p1 and err < 1e-6
    assert [r.speculation for r in res] == ["repaired", "repaired", "repaired", "hit"]
    print("multi-rank speculative run_stream: OK")
dist.destroy_process_group()
