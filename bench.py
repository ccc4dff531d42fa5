#!/usr/bin/env python3
"""bench.py -- spectra/s of the xmris `.xmr` hot path on MI355X.

One "step" = one full pass of  zero_fill -> apodize_exp -> to_spectrum -> autophase  over one
synthetic batch that is already resident in HBM, run by the library's streaming executor
`xmris_amd.pipeline.run_stream`.  Default (speculative) schedule, per step:
    guess stage: coarse spectra of every row (`xm_guess_rows`: the first 512 windowed samples on a 1024-bin grid,
    one wave per row, on the matrix cores: `k_coarse_mfma`) -> exact transform of every row whose estimate lies
    within a band of the largest (`xm_guess_refine`, branch and bound; the last workgroup writes the winning FID as
    complex128 into pinned host memory)
 -> that ONE row's spectrum computed by the host with the reference's numpy statements (`pipeline.winner_spectrum`:
    the slice the search runs on is the reference's bit for bit)
 -> the (p0, p1) search on it: scipy's differential evolution restated natively, on native host threads of the
    library (`xm_hostsearch_submit`) or, where the host is short of cores, as one workgroup on a reserved CU
    (`xm_search_launch`); the polish follows the reference's route wherever it iterates
 -> main kernel (fused zero-fill + window + FFT + fftshift + phase ramp in closed form + global arg-max key; reads
    the FIDs, writes the spectra) -> the TRUE global arg-max compared with the guess before the output buffer is
    reused; a wrong guess is repaired exactly (`speculation`, `speculation_miss` in the JSON line)
`--no-speculate`: the classic schedule -- an arg-max pre-pass (fused zero-fill + window + FFT + |X|^2 maxima,
nothing written) finds the winning row before the search, the main kernel only writes.
Over ranks: one O(1) exchange of (max, global flat index) per decision and one broadcast of (p0, p1).
Nothing is skipped or cached between steps.  Steps are independent datasets, so by default the guess (or
pre-pass) kernel of step i+1 is queued while the host solves step i (`--no-overlap` serialises them).
Workload at N=1: BASELINE.json configs[2] (65,536 voxels x 4096-pt complex64 FIDs zero-filled to 8192).  With
--gpus N every rank owns its own 65,536-voxel shard of ONE dataset (weak scaling).

`--gpus N` without a launcher (no WORLD_SIZE in the environment): this process starts N rank processes itself, one
per GPU, BEFORE anything touches a GPU (it never re-executes a process that has), relays rank 0's JSON line and exits
with the worst child status; it refuses to run when fewer than N devices are visible.

Prints ONE JSON line on rank 0's stdout (see the driver contract; library chatter goes to stderr); `roofline`
prices the dominant (main) kernel by HIP events on its stream, `cpu_baseline` times the CPU oracle.  The line
carries its own footnotes, measured after the timed region (they never enter `value`): the classic schedule's
rate, one dataset end to end without cross-dataset overlap, the cost of a wrong guess, a complex128 sub-record,
BASELINE configs[1] and configs[4] end to end (`configs`), the README quick start against the oracle (`parity`);
with --gpus N: `per_rank` (every rank's device period, search latency, searches owned).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# HBM traffic of ONE launch of the main kernel comes from the rocprofv3 counter passes of the SAME kernel on the SAME
# workload, kept as a small JSON next to their text output (scripts/collect_profiles.sh -> scripts/pmc_json.py:
# separate --pmc runs; FETCH_SIZE x 2 -- gfx950 reports a wide coalesced streaming read at half its bytes,
# MI355X_MICROARCH.md section HBM -- plus WRITE_SIZE, exact for 16-byte streaming stores).  bench.py reads that file and
# refuses the number (traffic: null) when the kernel it launches or the workload differs from what was profiled: a
# figure from another kernel cannot go stale silently.
PMC_FILES = {"c64": "profiles/r04/pmc_main_kernel.json", "c128": "profiles/r04/pmc_c128_main.json"}


def pmc_traffic(dtype_key, kernel_name, nv, nt, N):
    """(bytes per launch or None, provenance string) for `kernel_name` on nv x nt -> N from the committed counter file."""
    path = os.path.join(ROOT, PMC_FILES[dtype_key])
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, f"no counter file ({PMC_FILES[dtype_key]})"
    want = kernel_name.split(" (")[0].replace(" ", "")
    have = str(rec.get("kernel", "")).replace(" ", "")
    if have != want:
        return None, f"{PMC_FILES[dtype_key]} profiles {rec.get('kernel')!r}, this run launches {kernel_name!r}"
    if [rec.get("voxels"), rec.get("n_time"), rec.get("target_points")] != [nv, nt, N]:
        return None, f"{PMC_FILES[dtype_key]} was collected on another workload"
    total = int((2.0 * float(rec["FETCH_SIZE_KB"]) + float(rec["WRITE_SIZE_KB"])) * 1024)
    return total, (f"rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE of this kernel on this workload, "
                   f"{PMC_FILES[dtype_key]} (commit {rec.get('commit', '?')}); not re-measured in this run")


def synth_fids(torch, n_voxel, n_time, dt, voxel_offset, n_voxel_total, device, dtype, seed=42, star=None):
    """three damped lines + complex noise, per-voxel amplitude, one designated
    brightest voxel (unique global maximum; `star`, default n_voxel_total // 3).  Generated on the device."""
    t = np.arange(n_time) * dt
    amps, damps, freqs = (1.0, 0.5, 0.3), (20.0, 33.0, 25.0), (300.0, -800.0, 1100.0)
    base = sum(a * np.exp(-d * t) * np.exp(2j * np.pi * f * t) for a, d, f in zip(amps, damps, freqs))
    base_d = torch.from_numpy(base).to(device=device, dtype=dtype)
    v = torch.arange(voxel_offset, voxel_offset + n_voxel, device=device, dtype=torch.float64)
    amp = 0.5 + torch.remainder(v, 997.0) / 997.0
    star = n_voxel_total // 3 if star is None else int(star)
    amp = torch.where(v == float(star), torch.full_like(amp, 2.0), amp)
    rd = torch.float32 if dtype == torch.complex64 else torch.float64
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + voxel_offset)
    x = torch.empty((n_voxel, n_time), dtype=dtype, device=device)
    chunk = 8192
    for s in range(0, n_voxel, chunk):
        e = min(n_voxel, s + chunk)
        noise = torch.randn((e - s, n_time, 2), generator=gen, device=device, dtype=rd) * (0.02 / np.sqrt(2.0))
        x[s:e] = amp[s:e, None].to(rd) * base_d[None, :] + torch.view_as_complex(noise)
    return x, t


def synth_hetero(torch, n_voxel, n_time, dt, seed, device, dtype):
    """A heterogeneous dataset family for the speculative schedule's hit rate (nothing like SURVEY 8(d)'s one spectral
    shape for every voxel): per voxel 1-8 lines with random amplitudes (0.2-1 x a voxel gain of 0.5-1.5), widths
    2-60 Hz (FWHM), frequencies +-2 kHz and phases; 5 % noise-only voxels; complex noise of sigma 0.02; and ONE
    lipid-like voxel -- eight broad (50 Hz) lines of amplitude 3, 90 Hz apart -- whose windowed L1 norm is the largest
    of the dataset while its peak is not.  Returns (x [n_voxel, n_time], t, lipid_row).  The numpy twin used by the
    offline study is scripts/study_guess_statistics.py::hetero_numpy."""
    gen = torch.Generator(device=device)
    gen.manual_seed(7919 * (seed + 1))
    rd = torch.float32 if dtype == torch.complex64 else torch.float64
    t = np.arange(n_time) * dt
    td = torch.from_numpy(t).to(device=device, dtype=torch.float64)
    x = torch.empty((n_voxel, n_time), dtype=dtype, device=device)

    def uni(n, lo, hi):
        return lo + (hi - lo) * torch.rand(n, generator=gen, device=device, dtype=torch.float64)

    chunk = 4096
    for s in range(0, n_voxel, chunk):
        m = min(n_voxel, s + chunk) - s
        n_lines = torch.randint(1, 9, (m,), generator=gen, device=device)
        gain = uni(m, 0.5, 1.5)
        silent = torch.rand(m, generator=gen, device=device, dtype=torch.float64) < 0.05
        acc = torch.zeros((m, n_time), dtype=torch.complex128, device=device)
        for j in range(8):
            on = ((n_lines > j) & ~silent).to(torch.float64)
            a = uni(m, 0.2, 1.0) * gain * on
            w, f, ph = uni(m, 2.0, 60.0), uni(m, -2000.0, 2000.0), uni(m, 0.0, 2 * np.pi)
            rate = torch.complex(-np.pi * w, 2 * np.pi * f)
            acc += (a * torch.exp(1j * ph))[:, None] * torch.exp(rate[:, None] * td[None, :])
        noise = torch.randn((m, n_time, 2), generator=gen, device=device, dtype=torch.float64) * (0.02 / np.sqrt(2.0))
        x[s:s + m] = (acc + torch.view_as_complex(noise)).to(dtype)
    lipid = int(torch.randint(0, n_voxel, (1,), generator=gen, device=device).item())
    row = torch.zeros(n_time, dtype=torch.complex128, device=device)
    for j in range(8):
        row += 3.0 * torch.exp(torch.complex(torch.tensor(-np.pi * 50.0), torch.tensor(2 * np.pi * (-600.0 + 90.0 * j))).to(device) * td)
    x[lipid] = row.to(dtype)
    return x, t, lipid


def self_launch(args, argv):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as child processes (the environment torchrun
    would give them), relay rank 0's stdout.  Nothing here initialises a GPU."""
    import socket
    import subprocess

    import torch  # importing is safe; device_count() does not create a context on this stack

    n_dev = torch.cuda.device_count()
    if n_dev < 1 or (n_dev < args.gpus and not args.share_gpu):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but {n_dev} device(s) visible\n")
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), XM_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0]
    rcs = [p.wait() for p in procs]
    sys.stdout.buffer.write(out0)
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--voxels", type=int, default=65536, help="voxels per GPU")
    ap.add_argument("--n-time", type=int, default=4096)
    ap.add_argument("--target-points", type=int, default=8192)
    ap.add_argument("--lb", type=float, default=5.0)
    ap.add_argument("--dtype", choices=["c64", "c128"], default="c64")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="do not queue the next step's pre-pass while the host solves the current one")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dist-backend", default="nccl",
                    help="rehearsal only: 'gloo' lets several ranks share ONE GPU (with --share-gpu) to exercise "
                         "the multi-rank code path on a single-GPU box; the driver uses the default (RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--no-speculate", action="store_true",
                    help="classic schedule: arg-max pre-pass (an FFT of every FID) instead of the verified guess of the "
                         "winning row from the FIDs' windowed L1 norms (pipeline.run_stream(speculate=...))")
    ap.add_argument("--exchange", choices=["shm", "gloo"], default="shm",
                    help="N > 1: the O(1) host exchange goes through a shared-memory page (one node) or gloo")
    ap.add_argument("--prime-ms", type=float, default=200.0,
                    help="untimed steps run before the warm-up until this much wall time has passed (clocks, caches, "
                         "page tables and the software pipeline reach their steady state however short --warmup is)")
    ap.add_argument("--datasets", type=int, default=4,
                    help="distinct synthetic datasets the steps rotate through (different noise, different brightest voxel)")
    ap.add_argument("--hetero-sets", type=int, default=64,
                    help="footnote: datasets of the heterogeneous family (synth_hetero) the hit rate is measured on")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` record (BASELINE configs[1] and [4] end to end)")
    ap.add_argument("--only-configs", action="store_true", help="print the `configs` record alone (experiments)")
    ap.add_argument("--no-footnotes", action="store_true",
                    help="skip the extra measurements after the timed region (classic schedule, single dataset, forced "
                         "miss, complex128)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, sys.argv[1:]))

    # the all-cores CPU baseline's worker processes are forked NOW, before anything can have touched a GPU
    cpu_pool = None
    if args.gpus == 1 and "WORLD_SIZE" not in os.environ and not args.no_cpu_baseline:
        try:
            dt0 = 1.0 / 5000.0
            cpu_pool = CpuPool(min(16384, args.voxels), args.n_time, np.arange(args.n_time) * dt0, args.target_points, args.lb)
            import atexit

            atexit.register(cpu_pool.close)  # (whatever happens below: the workers and the shared block go away)
        except Exception as e:  # noqa: BLE001
            print(f"all-cores CPU baseline unavailable: {e!r}", file=sys.stderr)

    # Libraries chat on stdout (RCCL prints a five-line banner at communicator creation, gloo a line per rank): the
    # contract is ONE JSON line there.  File descriptor 1 is pointed at stderr for the whole run; the result line goes
    # to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    from xmris_amd import autophase_solver as aps
    from xmris_amd import device as dev
    from xmris_amd import pipeline, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if args.share_gpu:
        local_rank = 0
    n_dev = torch.cuda.device_count()
    if not args.share_gpu and int(os.environ.get("LOCAL_WORLD_SIZE", world)) > n_dev > 1:
        raise SystemExit(f"{world} ranks on this node but only {n_dev} devices visible")
    local_rank %= max(1, n_dev)  # a launcher may expose one device per rank
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if os.environ.get("XM_BENCH_WATCHDOG"):  # debugging aid: dump every thread's stack and exit after so many seconds
        import faulthandler

        faulthandler.dump_traceback_later(float(os.environ["XM_BENCH_WATCHDOG"]), exit=True)
    if args.only_configs:
        os.write(result_fd, (json.dumps({"configs": configs_note(torch, pipeline, device, args)}) + "\n").encode())
        return
    dist = None
    host_group = None
    shm = None
    rccl_ranks = None
    rccl_group = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # single node: the container hostname may not resolve
        # The default group is gloo: the O(1) arg-max exchange and (p0, p1) broadcast are host-side metadata and must
        # not queue behind the kernels of other datasets already on the GPU stream.  RCCL carries the barrier and the
        # max-over-ranks timing in a group of its own; should it fail to come up (every rank learns of it through
        # gloo), the run goes on with gloo alone and says so in the JSON line instead of dying without a number.
        dist.init_process_group("gloo")
        host_group = dist.group.WORLD
        if args.dist_backend == "nccl":
            def all_ok(ok):  # every rank learns through gloo whether EVERY rank got this far
                flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return bool(flag.item())

            # step by step, agreeing through gloo BEFORE the first RCCL collective: a rank whose group creation failed
            # would otherwise leave the others waiting inside RCCL until its watchdog fires
            try:
                rccl_group = dist.new_group(backend="nccl", device_id=device)
                made = True
            except Exception as e:  # noqa: BLE001 -- anything RCCL throws at start-up
                print(f"[rank {rank}] RCCL group unavailable ({e!r}); barrier and timing go through gloo", file=sys.stderr)
                made = False
            if not all_ok(made):
                rccl_group = None
            else:
                try:
                    ones = torch.ones(1, dtype=torch.int32, device=device)
                    dist.all_reduce(ones, group=rccl_group)  # how many ranks RCCL really spans
                    rccl_ranks = int(ones.item())
                    worked = True
                except Exception as e:  # noqa: BLE001
                    print(f"[rank {rank}] RCCL all-reduce failed ({e!r}); barrier and timing go through gloo", file=sys.stderr)
                    worked = False
                if not all_ok(worked):
                    rccl_group, rccl_ranks = None, None
        # on one node the same exchange goes through a shared-memory page (microseconds instead of two
        # loopback collectives per dataset); every rank must agree on the choice, so failures are gathered
        if args.exchange == "shm":
            try:
                shm = sharding.ShmExchange.create(dist, group=host_group)
            except OSError as e:  # raised on every rank alike: all of them fall back to gloo
                print(f"[rank {rank}] shared-memory exchange unavailable ({e}); using gloo", file=sys.stderr)

    cdtype = torch.complex64 if args.dtype == "c64" else torch.complex128
    rdtype = torch.float32 if args.dtype == "c64" else torch.float64
    bytes_per = 8 if args.dtype == "c64" else 16
    nv, nt, N = args.voxels, args.n_time, args.target_points
    dt = 1.0 / 5000.0
    # a ring of distinct datasets (SURVEY 8(d)'s recipe: own noise each, the brightest voxel somewhere else each time,
    # on another rank when there are several); the steps rotate through it
    n_total = world * nv
    xs = []
    for k in range(max(1, args.datasets)):
        xk, t = synth_fids(torch, nv, nt, dt, rank * nv, n_total, device, cdtype, seed=42 + 1009 * k,
                           star=(n_total // 3 + k * (n_total // 5) + 7 * k) % n_total)
        xs.append(xk)
    x = xs[0]

    # host metadata exactly as the accessor layer computes it (fid.py:257-263, 136; fourier.py:95-98, 31)
    tt = t[0] + np.arange(N) * (t[1] - t[0])
    freq = np.roll(np.fft.fftfreq(N, d=tt[1] - tt[0]), N // 2)

    # two output buffers, used alternately: a streaming caller's datasets have outputs of their own, and the
    # verification of step i then need not finish before step i+1's main pass is queued
    out = torch.empty((nv, N), dtype=cdtype, device=device)
    out_b = torch.empty((nv, N), dtype=cdtype, device=device)
    times = {"pre_ms": [], "sel_ms": [], "main_ms": [], "solve_ms": [], "exchange_ms": [], "gen_ms": [], "polish_ms": [], "table_ms": [], "period_ms": []}
    last = {}

    plan = pipeline.make_plan(x, t, N, args.lb)
    assert np.array_equal(plan.freq, freq)
    ddev = "cpu"
    overlap = not args.no_overlap
    speculate = not args.no_speculate
    spec_stats = {}

    # O(1) per dataset: (max, global flat index) per rank -> the winner on every rank; (p0, p1) from its owner
    def exchange(amax, gflat):
        if shm is not None:
            owner, gwin, _ = shm.exchange_argmax(amax, gflat)
        else:
            owner, gwin, _ = sharding.exchange_argmax(amax, gflat, dist, ddev, group=host_group)
        return owner == rank, gwin, owner

    def broadcast(values, owner):
        if shm is not None:
            return shm.broadcast_params(values, owner)
        return sharding.broadcast_params(values, owner, dist, ddev, group=host_group)

    # (XM_BENCH_SOLO_EXCHANGE=1: one rank through the executor's MULTI-rank code path -- trivial exchange / broadcast
    # callables -- to measure what that path's fixed call order costs on a GPU of its own; a rehearsal switch)
    solo_exchange = solo_broadcast = None
    if world == 1 and os.environ.get("XM_BENCH_SOLO_EXCHANGE"):
        solo_exchange = lambda amax, gflat: (True, gflat, 0)  # noqa: E731
        solo_broadcast = lambda values, owner: [float(v) for v in values]  # noqa: E731

    def run_steps(n_steps, record):
        """n_steps complete passes of the hot path = xmris_amd.pipeline.run_stream over n_steps independent
        datasets (the library's software-pipelined executor: with `overlap` the device runs the pre-pass of
        dataset i+1 and the main pass of dataset i-1 while the host searches (p0, p1) for dataset i; every
        step does all of its own work inside this call, nothing is left over or reused).  The steps rotate
        through the ring of distinct datasets; outputs alternate between two buffers."""
        trace = []
        results = pipeline.run_stream([xs[k % len(xs)] for k in range(n_steps)],
                                      [out if k % 2 == 0 else out_b for k in range(n_steps)], plan,
                                      exchange=exchange if world > 1 else solo_exchange,
                                      broadcast=broadcast if world > 1 else solo_broadcast, rank_offset_rows=rank * nv,
                                      overlap=overlap, trace=trace, speculate=speculate)
        for r in results:
            if speculate:
                spec_stats[r.speculation] = spec_stats.get(r.speculation, 0) + (1 if record else 0)
                if getattr(r, "hedged", False) and record:
                    spec_stats["hedged"] = spec_stats.get("hedged", 0) + 1
                if record and r.mine and isinstance(getattr(r, "timing", None), dict):  # which engine searched it
                    eng_key = "searches_device" if r.timing.get("device") else "searches_host"
                    spec_stats[eng_key] = spec_stats.get(eng_key, 0) + 1
        res = results[-1]
        last.update(p0=res.p0, p1=res.p1, pivot=res.pivot, flat=res.flat_index, owner=res.owner)
        for r in results:
            if r.mine:
                last["nfev"] = r.nfev
        return trace, results

    def read_times(trace, results):
        """The HIP-event durations of a finished run_steps: read AFTER the wall clock has stopped (sixty elapsed_time
        queries and the bookkeeping around them are measurement, not work -- round 2 had them inside the timed region)."""
        n_steps = len(results)
        torch.cuda.synchronize()
        for i, (e, r) in enumerate(zip(trace, results)):
            times["exchange_ms"].append((e["t_exchanged"] - e["t_start"]) * 1e3)
            times["solve_ms"].append((e["t_solved"] - e["t_exchanged"]) * 1e3)
            times["table_ms"].append((e["t_table"] - e["t_solved"]) * 1e3)
            if r.mine:
                times["gen_ms"].append(r.timing.get("generations_ms", 0.0))
                times["polish_ms"].append(r.timing.get("polish_ms", 0.0))
            times["pre_ms"].append(e["pre0"].elapsed_time(e["pre1"]))
            if "sel1" in e:  # speculative schedule: exact check of the candidates + the winner's fp64 spectrum
                times["sel_ms"].append(e["pre1"].elapsed_time(e["sel1"]))
            times["main_ms"].append(e["main0"].elapsed_time(e["main1"]))
            if i + 1 < n_steps:  # device-side period: start of main pass i -> start of main pass i+1
                times["period_ms"].append(e["main0"].elapsed_time(trace[i + 1]["main0"]))

    def barrier():
        if dist is not None:
            if rccl_group is not None:
                dist.barrier(group=rccl_group, device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # A full (generation-2) garbage collection over the interpreter's ~10^5 long-lived objects (torch, numpy) takes
    # 10+ ms -- several steps.  Collect now and move everything alive into the permanent generation: the cyclic
    # garbage the timed loop creates is then collected in microseconds.  Done BEFORE the priming so that the device
    # does not sit idle (and clock down) between the last untimed step and the first timed one.
    import gc

    run_steps(2, False)  # first use: plans, tables, pinned buffers, worker threads
    gc.collect()
    gc.freeze()
    prime_spent_ms = 0.0
    if args.prime_ms > 0:  # untimed: until the wall clock says the device has been busy for a while
        t_prime = time.perf_counter()
        best, last_batch = float("inf"), float("inf")
        while True:
            # every run_steps is a sequence of exchanges: all ranks must make the SAME number of calls, so rank 0's
            # clock decides for everyone (each rank reading its own clock left them one call apart -- a deadlock).
            # Past prime_ms the priming goes on (to at most 6 x prime_ms) while the last batch was still more than 4 %
            # slower than the best one seen: a first process on a fresh box (clocks, page cache, thread pools) settles
            # within the untimed part instead of inside the K timed steps.
            spent = (time.perf_counter() - t_prime) * 1e3
            go = spent < args.prime_ms or (last_batch > 1.04 * best and spent < 6 * args.prime_ms)
            if dist is not None:
                flag = torch.tensor([1 if go else 0], dtype=torch.int32)
                dist.broadcast(flag, src=0, group=host_group)
                go = bool(flag.item())
            if not go:
                prime_spent_ms = spent
                break
            t_b = time.perf_counter()
            run_steps(8, False)
            torch.cuda.synchronize()
            last_batch = time.perf_counter() - t_b
            best = min(best, last_batch)
    if args.warmup:
        run_steps(args.warmup, False)
    barrier()
    import resource

    def cgroup_cpu_stat():
        try:
            return {k: int(v) for k, v in (ln.split() for ln in open("/sys/fs/cgroup/cpu.stat"))}
        except (OSError, ValueError):
            return {}

    from xmris_amd import _lib as _xl

    backups0 = _xl.load().xm_solver_pool_backups()
    cg0 = cgroup_cpu_stat()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    cpu0 = sum(os.times()[:2])
    t_start = time.perf_counter()
    timed_trace, timed_results = run_steps(args.steps, True)
    barrier()
    elapsed = time.perf_counter() - t_start
    cpu_cores = (sum(os.times()[:2]) - cpu0) / elapsed  # this process's threads, in cores, over the timed region
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    host_noise = {"major_faults": ru1.ru_majflt - ru0.ru_majflt, "minor_faults": ru1.ru_minflt - ru0.ru_minflt,
                  "involuntary_switches": ru1.ru_nivcsw - ru0.ru_nivcsw}  # of this process, over the timed region
    host_noise["solver_shares_backed_up"] = _xl.load().xm_solver_pool_backups() - backups0  # xm_solver_obj.cpp, "Stragglers"
    cg1 = cgroup_cpu_stat()
    for k in ("nr_throttled", "throttled_usec", "nr_periods"):  # CPU-bandwidth throttling of the whole cgroup
        if k in cg0 and k in cg1:
            host_noise["cgroup_" + k] = cg1[k] - cg0[k]
    read_times(timed_trace, timed_results)
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if rccl_group is not None else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=rccl_group)
        elapsed = float(tmax.item())

    if os.environ.get("XM_BENCH_DEBUG"):  # per-step device times of the timed region (stderr)
        print("main_ms", [round(v, 3) for v in times["main_ms"]], file=sys.stderr)
        print("period_ms", [round(v, 3) for v in times["period_ms"]], file=sys.stderr)
        print("pre_ms", [round(v, 3) for v in times["pre_ms"]], file=sys.stderr)
        for k in ("exchange_ms", "solve_ms", "gen_ms", "polish_ms", "table_ms"):
            print(k, [round(v, 3) for v in times[k]], file=sys.stderr)
        if timed_trace and "t_search_begin" in timed_trace[-1]:  # where a late search lost its time
            t0 = timed_trace[0]["t_start"]
            for name, a, b in (("search_dispatch_ms", "t_exchanged", "t_search_begin"), ("search_run_ms", "t_search_begin", "t_search_end"),
                               ("search_done_to_collect_ms", "t_search_end", "t_collect"), ("collect_to_solved_ms", "t_collect", "t_solved")):
                print(name, [round((e[b] - e[a]) * 1e3, 3) if a in e and b in e else None for e in timed_trace], file=sys.stderr)
            print("t_collect_ms", [round((e["t_collect"] - t0) * 1e3, 2) for e in timed_trace if "t_collect" in e], file=sys.stderr)
        if timed_trace and "t_selected" in timed_trace[-1]:  # one line per rank: where the launch thread was when (ms)
            t0 = timed_trace[0]["t_start"]
            marks = ("t_start", "t_selected", "t_exchanged", "t_collect", "t_solved")
            if "t_call" in timed_trace[0]:
                tc, last = timed_trace[0]["t_call"], timed_trace[-1]
                print(f"[rank {rank}] timed region {elapsed * 1e3:.2f} ms: executor entered at {(tc - t_start) * 1e3:.2f}, first search "
                      f"started at {(t0 - t_start) * 1e3:.2f}, last main pass queued at {(last.get('t_last_queued', tc) - t_start) * 1e3:.2f}, "
                      f"executor returned at {(last.get('t_return', tc) - t_start) * 1e3:.2f}", file=sys.stderr)
            sys.stderr.flush()
            os.write(2, (f"\n[rank {rank}] timeline (start, selected, exchanged, collect, solved) " + " | ".join(
                " ".join(f"{(e[m] - t0) * 1e3:.1f}" if m in e else "-" for m in marks) for e in timed_trace) + " |\n").encode())  # (one write: the ranks share stderr)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * nv * args.steps / elapsed
    main_ms = float(np.mean(times["main_ms"]))
    pre_ms = float(np.mean(times["pre_ms"]))
    alg_bytes = (bytes_per * nt + bytes_per * N) * nv  # read each FID once + write each spectrum once
    achieved = alg_bytes / (main_ms * 1e-3) / 1e9
    sel_ms = float(np.mean(times["sel_ms"])) if times["sel_ms"] else 0.0
    stream_ms = main_ms + pre_ms + sel_ms  # every kernel of a step: guess (or pre-pass) + selection stage + main pass
    # the kernels are named by the dispatcher that launched them (`xm_last_kernel_string`, read right behind the launch
    # by the executor), not by this script
    main_name = next((e["main_kernel"] for e in reversed(timed_trace) if e.get("main_kernel")), "xm_pipeline_fused_ramp main pass")
    guess_name = next((e["guess_kernel"] for e in reversed(timed_trace) if e.get("guess_kernel")), None)
    kernel = main_name + " (zero-fill+window+FFT+fftshift+phase" + ("+global arg-max)" if speculate else ")")
    traffic, traffic_source = pmc_traffic(args.dtype, kernel, nv, nt, N)

    result = {
        "metric": "spectra/sec (zero_fill->apodize->FFT->autophase), n_time=4096; HBM-roofline %",
        "value": value,
        "unit": "spectra/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.dtype == "c64" else "f64",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[2]: {nv} voxels x {nt}-pt complex FID -> zero-fill {N}, lb={args.lb}, "
                        f"autophase(acme, single), storage {args.dtype}, per GPU",
            "voxels_per_gpu": nv, "n_time": nt, "target_points": N, "parallelism": f"voxel-shard x{world}",
            "distinct_datasets": len(xs),
        },
        "roofline": {
            "bound": "hbm", "kernel": kernel,
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": main_ms,
        },
        "breakdown_ms": {
            ("guess_kernel" if speculate else "prepass_kernel"): pre_ms, "guess_kernel_name": guess_name,
            "selection_stage_kernels": sel_ms if speculate else None,  # exact check of the candidates + the winner's fp64 spectrum
            "main_kernel": main_ms,
            "selection_wait_and_exchange": float(np.mean(times["exchange_ms"])),
            # speculative schedule: searches run three datasets ahead on worker threads -- this is the LATENCY from a
            # dataset's exchange to the moment its (p0, p1) is consumed, not host time on the critical path
            ("search_latency_exchange_to_use" if speculate else "slice_de_solve_broadcast"): float(np.mean(times["solve_ms"])),
            "solver_generations": float(np.mean(times["gen_ms"])) if times["gen_ms"] else None,
            "solver_polish": float(np.mean(times["polish_ms"])) if times["polish_ms"] else None,
            "device_period_min_median_max": ([float(np.min(times["period_ms"])), float(np.median(times["period_ms"])),
                                              float(np.max(times["period_ms"]))] if times["period_ms"] else None),
            "device_period_p90_p99": ([float(np.percentile(times["period_ms"], 90)),
                                       float(np.percentile(times["period_ms"], 99))] if times["period_ms"] else None),
            "streaming_spectra_per_s_per_gpu": nv / (stream_ms * 1e-3),
            "streaming_roofline_frac": alg_bytes / (stream_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        },
        "end_to_end_roofline_frac": value / world * (bytes_per * nt + bytes_per * N) / 1e9 / HBM_PEAK_GBPS,
        "autophase": {k: last.get(k) for k in ("p0", "p1", "pivot", "flat", "owner", "nfev")},
        "schedule": (("guess kernels run 4 and (p0, p1) searches 3 datasets ahead of the main pass being queued (independent "
                      "datasets); the main pass returns the true global arg-max and every guess is verified (repaired if "
                      "wrong) before the next main pass" if speculate else
                      "pre-pass of step i+1 overlaps the host solve of step i (independent datasets)") if overlap
                     else "strictly serial steps"),
        "speculation": ({"enabled": True, "hit": spec_stats.get("hit", 0), "repaired": spec_stats.get("repaired", 0),
                         "searches_started_twice": spec_stats.get("hedged", 0),
                         "searches_host_engine": spec_stats.get("searches_host", 0),
                         "searches_device_engine": spec_stats.get("searches_device", 0)}
                        if speculate else {"enabled": False}),
        "prime_ms": args.prime_ms,
        "prime_ms_spent": round(prime_spent_ms, 1),
        "host_noise_timed_region": host_noise,
        "host_cores_used_rank0": cpu_cores,
    }
    if world > 1:
        result["rccl_ranks"] = rccl_ranks  # None: the barrier / timing reduction ran on gloo (see stderr)
        # how every rank's host kept up, in the line itself: a scaling record explains itself
        mine_rec = {"rank": rank, "device_period_median_ms": float(np.median(times["period_ms"])) if times["period_ms"] else None,
                    "search_latency_mean_ms": float(np.mean(times["solve_ms"])), "search_latency_max_ms": float(np.max(times["solve_ms"])),
                    "searches_owned": len(times["gen_ms"]),
                    "search_generations_mean_ms": float(np.mean(times["gen_ms"])) if times["gen_ms"] else None,
                    "main_kernel_ms": main_ms, "host_cores_used": cpu_cores, "ms_per_step_local": ms_per_step,
                    "hedged": spec_stats.get("hedged", 0), "repaired": spec_stats.get("repaired", 0),
                    "searches_device_engine": spec_stats.get("searches_device", 0)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine_rec, group=host_group)
        result["per_rank"] = gathered
        # every rank reports how its host kept up (stderr): the search latency must stay below the look-ahead (two
        # device periods) or the searches pace the steps
        per = times["period_ms"]
        print(f"[rank {rank}] search_latency_exchange_to_use {np.mean(times['solve_ms']):.3f} ms (max {np.max(times['solve_ms']):.3f}), "
              f"device_period median {np.median(per) if per else float('nan'):.3f} ms, searches owned "
              f"{len(times['gen_ms'])}/{args.steps}, generations {np.mean(times['gen_ms']) if times['gen_ms'] else float('nan'):.3f} ms, "
              f"team budget {aps.stream_threads()} threads on {len(os.sched_getaffinity(0))} CPUs, this process used "
              f"{cpu_cores:.2f} cores, ms/step {ms_per_step:.3f}",
              file=sys.stderr)
    if not args.no_footnotes and world == 1:
        result.update(footnotes(torch, pipeline, dev, xs, t, (out, out_b), plan, N, args, speculate, main_ms, alg_bytes))
        if speculate and "heterogeneous" in result:
            result["speculation"]["hit_rate_heterogeneous"] = result["heterogeneous"]["hit_rate"]
        if speculate and "speculation_miss" in result:  # the hit counts above are the timed region's; this is the price
            result["speculation"]["miss_penalty_ms"] = result["speculation_miss"]["miss_penalty_ms"]

    if not args.no_footnotes and not args.no_configs and world == 1:
        try:
            result["configs"] = configs_note(torch, pipeline, device, args)
        except Exception as e:  # noqa: BLE001 -- a footnote must not cost the headline
            result["configs"] = {"error": repr(e)[:200]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(x, t, N, args.lb, args.cpu_seconds, nv)
        try:  # (same leg: the oracle as the checker)
            result["parity"] = parity_note(torch, pipeline, dev, device)
        except Exception as e:  # noqa: BLE001
            result["parity"] = {"error": repr(e)[:200]}
        if cpu_pool is not None:
            try:
                sample = x[:cpu_pool.shape[0]].to(torch.complex64).cpu().numpy()
                result["cpu_baseline_all_cores"] = cpu_pool.run(sample, nv, result["cpu_baseline"]["de_solve_s"])
            except Exception as e:  # noqa: BLE001 -- the headline must not die with a worker
                result["cpu_baseline_all_cores"] = {"error": repr(e)[:200]}
            cpu_pool.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


def footnotes(torch, pipeline, dev, xs, t, outs, plan, N, args, speculate, main_ms, alg_bytes):
    """What the headline does not say by itself, measured after the timed region on the same buffers:
      value_classic_schedule   the same workload with the arg-max pre-pass instead of the verified guess
      single_dataset_ms        ONE dataset end to end (guess -> search -> main -> verify), nothing to overlap with
      speculation_miss         the price of a wrong guess: one row gets a late burst (the tallest peak of its dataset,
                               in samples the guess stage does not read) and every step has to be repaired
      heterogeneous            the speculative schedule on `--hetero-sets` DISTINCT datasets of the heterogeneous
                               family (synth_hetero: 1-8 lines per voxel, widths 2-60 Hz, noise-only voxels, a
                               lipid-like voxel with the largest L1 norm): measured hits / steps, the measured rate
                               including whatever repairs happened, and the expected rate from the hit rate
      c128                     complex128 storage (the reference's arithmetic, fid.py:136-139), same voxel count"""
    import time

    x = xs[0]
    out = outs[0]
    nv = x.shape[0]
    notes = {}
    ring = lambda n: [xs[k % len(xs)] for k in range(n)]  # noqa: E731
    alt = lambda n: [outs[k % 2] for k in range(n)]  # noqa: E731

    def rate(inputs, outputs, pl, n, spec):
        pipeline.run_stream(inputs[:4], outputs[:4], pl, speculate=spec)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = pipeline.run_stream(inputs[:n], outputs[:n], pl, speculate=spec)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, res

    k = max(8, min(args.steps, 40))
    ms, _ = rate(ring(k), [out] * k, plan, k, False)
    notes["value_classic_schedule"] = nv / (ms * 1e-3)
    notes["classic_schedule_ms_per_step"] = ms
    singles = []
    for i in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipeline.run_stream([xs[i % len(xs)]], [out], plan, speculate=speculate)
        torch.cuda.synchronize()
        singles.append((time.perf_counter() - t0) * 1e3)
    notes["single_dataset_ms"] = float(np.median(singles[2:]))
    notes["single_dataset_spectra_per_s"] = nv / (notes["single_dataset_ms"] * 1e-3)
    if speculate:
        # forced miss: row 7 = a burst in samples 600...999 (the guess stage reads samples 0...511), the tallest peak
        xm = x.clone()
        tt = torch.from_numpy(np.asarray(t)).to(x.device)
        row = torch.zeros(x.shape[1], dtype=torch.complex128, device=x.device)
        lo, hi = min(600, x.shape[1] - 2), min(1000, x.shape[1])
        row[lo:hi] = 40.0 * torch.exp(2j * np.pi * 650.0 * tt[lo:hi])
        xm[7] = row.to(x.dtype)
        ms_miss, res = rate([xm] * 12, alt(12), plan, 12, True)
        ms_hit, _ = rate([x] * 12, alt(12), plan, 12, True)
        n_rep = sum(r.speculation == "repaired" for r in res)
        notes["speculation_miss"] = {"forced_miss_steps": 12, "repaired": n_rep, "ms_per_step_all_missed": ms_miss,
                                     "ms_per_step_all_hit_same_length": ms_hit,
                                     "miss_penalty_ms": (ms_miss - ms_hit) if n_rep == 12 else None}
        del xm
        try:
            notes["heterogeneous"] = hetero_note(torch, pipeline, x, t, outs, plan, args,
                                                 notes["speculation_miss"]["miss_penalty_ms"])
        except Exception as e:  # an out-of-memory box must not cost the headline
            notes["heterogeneous"] = {"error": repr(e)[:200]}
    if args.dtype == "c64":
        try:
            notes["c128"] = c128_note(torch, pipeline, xs, t, N, args, speculate)
        except Exception as e:  # an out-of-memory box must not cost the headline
            notes["c128"] = {"error": repr(e)[:200]}
    return notes


def hetero_note(torch, pipeline, x, t, outs, plan, args, miss_penalty_ms):
    """Hit rate of the speculative schedule on DISTINCT datasets of the heterogeneous family, in batches of 16
    resident datasets; every dataset is guessed, searched, transformed and verified once."""
    import time

    nv, nt = x.shape
    dt = float(t[1] - t[0])
    total = max(4, args.hetero_sets)
    per = min(16, total)
    hits = repaired = steps = 0
    wall = 0.0
    seed = 0
    while steps < total:
        n = min(per, total - steps)
        sets = []
        for _ in range(n):
            sets.append(synth_hetero(torch, nv, nt, dt, seed, x.device, x.dtype)[0])
            seed += 1
        if steps == 0:  # first use of these buffers: untimed (the headline has its warm-up too); results not counted
            pipeline.run_stream(sets[:2], [outs[0], outs[1]], plan, speculate=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = pipeline.run_stream(sets, [outs[k % 2] for k in range(n)], plan, speculate=True)
        torch.cuda.synchronize()
        wall += time.perf_counter() - t0
        hits += sum(r.speculation == "hit" for r in res)
        repaired += sum(r.speculation == "repaired" for r in res)
        steps += n
        del sets
    h = hits / steps
    ms = wall / steps * 1e3
    note = {"family": "synth_hetero: 1-8 lines/voxel, widths 2-60 Hz, 5% noise-only voxels, one lipid-like voxel "
                      "(largest L1 norm), distinct seeds", "datasets": steps, "hit": hits, "repaired": repaired,
            "hit_rate": h, "ms_per_step_measured": ms, "value_measured": nv / (ms * 1e-3)}
    if miss_penalty_ms is not None and repaired == 0:
        note["value_expected_at_hit_rate"] = note["value_measured"]
    elif miss_penalty_ms is not None:
        t_hit = ms - (1.0 - h) * miss_penalty_ms
        note["value_expected_at_hit_rate"] = nv / ((h * t_hit + (1.0 - h) * (t_hit + miss_penalty_ms)) * 1e-3)
    return note


def configs_note(torch, pipeline, device, args):
    """BASELINE configs[1] (32 x 32 x 16 voxels x 2048 -> 4096) and configs[4] (8 x 64 x 64 voxels x 1536, no zero fill:
    the 3 * 2^k plan) end to end through the SAME streaming executor as the headline, as records of their own: ms per
    dataset, the main kernel's time by HIP events, its fraction of the HBM peak and the end-to-end fraction of each
    config's own roofline (8 (n_time + n_out) bytes per spectrum) -- with the search on the device and on the host."""
    import os
    import time

    out = {}
    for name, nv, nt, N in (("configs[1] 16384 x 2048 -> 4096", 16384, 2048, 4096), ("configs[4] 32768 x 1536", 32768, 1536, 1536)):
        dt = 1.0 / 5000.0
        xs = []
        for k in range(4):
            xk, t = synth_fids(torch, nv, nt, dt, 0, nv, device, torch.complex64, seed=77 + 1009 * k,
                               star=(nv // 3 + k * (nv // 5) + 7 * k) % nv)
            xs.append(xk)
        outs = [torch.empty((nv, N), dtype=torch.complex64, device=device) for _ in range(2)]
        plan = pipeline.make_plan(xs[0], t, N, args.lb)
        bytes_per = 8 * (nt + N) * nv
        rec = {"voxels": nv, "n_time": nt, "n_out": N, "algorithmic_bytes_per_dataset": bytes_per, "dtype": "f32"}
        k2 = 60
        for engine in ("device", "host"):
            old = os.environ.get("XMRIS_AMD_SEARCH")
            os.environ["XMRIS_AMD_SEARCH"] = engine
            try:
                ins = [xs[k % 4] for k in range(k2)]
                ots = [outs[k % 2] for k in range(k2)]
                pipeline.run_stream(ins[:12], ots[:12], plan, speculate=True)
                torch.cuda.synchronize()
                best = None
                for _ in range(3):
                    trace = []
                    t0 = time.perf_counter()
                    res = pipeline.run_stream(ins, ots, plan, speculate=True, trace=trace)
                    torch.cuda.synchronize()
                    ms = (time.perf_counter() - t0) / k2 * 1e3
                    main = float(np.mean([e["main0"].elapsed_time(e["main1"]) for e in trace]))
                    cur = {"ms_per_dataset": ms, "spectra_per_s": nv / (ms * 1e-3), "main_kernel_ms": main,
                           "main_kernel_frac": bytes_per / (main * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                           "end_to_end_roofline_frac": bytes_per / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                           "hit": sum(r.speculation == "hit" for r in res), "repaired": sum(r.speculation == "repaired" for r in res),
                           "searches_on_device": sum(bool(r.timing.get("device")) for r in res)}
                    if best is None or cur["ms_per_dataset"] < best["ms_per_dataset"]:
                        best = cur
                rec["search_" + engine] = best
            except Exception as e:  # noqa: BLE001 -- a footnote must not cost the headline
                rec["search_" + engine] = {"error": repr(e)[:200]}
            finally:
                if old is None:
                    os.environ.pop("XMRIS_AMD_SEARCH", None)
                else:
                    os.environ["XMRIS_AMD_SEARCH"] = old
        out[name] = rec
        del xs, outs
    return out


def c128_note(torch, pipeline, xs, t, N, args, speculate):
    """complex128 storage -- the reference's arithmetic (fid.py:136-139 promotes to complex128) -- on the SAME voxel
    count as the headline, as a record of its own: rate, main-kernel time by HIP events, roofline fractions (16 B per
    sample: twice the bytes per spectrum)."""
    import time

    nv = xs[0].shape[0]
    k2 = 40
    x2 = [x.to(torch.complex128) for x in xs[:2]]
    # as the headline: outputs alternate between two buffers, the pipeline fill is inside the timed steps
    out2 = [torch.empty((nv, N), dtype=torch.complex128, device=xs[0].device) for _ in range(2)]
    plan2 = pipeline.make_plan(x2[0], t, N, args.lb)
    trace = []
    pipeline.run_stream([x2[k % 2] for k in range(8)], [out2[k % 2] for k in range(8)], plan2, speculate=speculate)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = pipeline.run_stream([x2[k % 2] for k in range(k2)], [out2[k % 2] for k in range(k2)], plan2, speculate=speculate,
                              trace=trace)
    torch.cuda.synchronize()
    ms2 = (time.perf_counter() - t0) / k2 * 1e3
    main2 = float(np.mean([e["main0"].elapsed_time(e["main1"]) for e in trace]))
    bytes2 = 16 * (x2[0].shape[1] + N) * nv
    main_name = next((e["main_kernel"] for e in reversed(trace) if e.get("main_kernel")), "xm_pipeline_fused_ramp main pass")
    kernel2 = main_name + " (zero-fill+window+FFT+fftshift+phase" + ("+global arg-max)" if speculate else ")")
    traffic2, source2 = pmc_traffic("c128", kernel2, nv, x2[0].shape[1], N)
    return {"voxels": nv, "steps": k2, "value": nv / (ms2 * 1e-3), "ms_per_step": ms2, "dtype": "f64",
            "roofline": {"bound": "hbm", "kernel": kernel2,
                         "achieved": bytes2 / (main2 * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": bytes2 / (main2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": bytes2,
                         "avg_launch_ms": main2, "traffic": traffic2, "traffic_source": source2},
            "main_kernel_ms": main2, "main_kernel_frac": bytes2 / (main2 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "end_to_end_roofline_frac": nv / (ms2 * 1e-3) * 16 * (x2[0].shape[1] + N) / 1e9 / HBM_PEAK_GBPS,
            "speculation": ({"hit": sum(r.speculation == "hit" for r in res),
                             "repaired": sum(r.speculation == "repaired" for r in res)} if speculate else None)}


def parity_note(torch, pipeline, dev, device):
    """BASELINE configs[0] (the README quick start: 5 x 1024 noise FIDs -> 2048, lb = 5, autophase) through the
    STREAMING executor this script times, against the CPU oracle (part of the cpu_baseline leg: the oracle is the
    checker).  north_star: <= 1e-5 of the spectrum's maximum.  On pure noise the landscape is flat -- and since round 4
    the search runs on the reference's slice bit for bit (`pipeline.winner_spectrum`), with scipy's generations and
    scipy's polish route: (p0, p1) are the oracle's (dp = 0) and the spectra sit at the storage floor.  The complex64
    record is compared with the oracle run on the SAME complex64 array (what the reference computes when handed it;
    `spectrum_rel_err_vs_complex128_data` is the older comparison, which also contains the rounding of the INPUT,
    amplified by the flat landscape)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xmris_oracle as orc

    rng = np.random.default_rng(42)
    t = np.linspace(0, 1, 1024)
    x = rng.standard_normal((5, 1024)) + 1j * rng.standard_normal((5, 1024))
    ref, info = orc.pipeline_values(x.astype(np.complex128), t, 2048, 5.0, peak_width=100)
    out = {"workload": "BASELINE configs[0]: 5 x 1024 noise FIDs -> 2048, lb=5, autophase, via run_stream(speculate=True)",
           "tolerance_north_star": 1e-5}
    ref128, info128 = ref, info
    for name, dt in (("c1_c128", torch.complex128), ("c1_c64", torch.complex64)):
        if dt == torch.complex64:
            ref, info = orc.pipeline_values(x.astype(np.complex64), t, 2048, 5.0, peak_width=100)
        xd = torch.from_numpy(x).to(device=device, dtype=dt)
        plan = pipeline.make_plan(xd, t, 2048, 5.0)
        outs = [torch.empty((5, 2048), dtype=dt, device=device) for _ in range(6)]
        res = pipeline.run_stream([xd] * 6, outs, plan, speculate=True)
        err = max(float(np.abs(o.cpu().numpy() - ref).max() / np.abs(ref).max()) for o in outs)
        out[name] = {"spectrum_rel_err": err, "dp0_deg": abs(res[-1].p0 - info["p0"]), "dp1_deg": abs(res[-1].p1 - info["p1"]),
                     "flat_index_equal": bool(res[-1].flat_index == info["flat_idx"]), "nfev": int(res[-1].nfev),
                     "nfev_oracle": int(info["nfev"])}
        if dt == torch.complex64:
            out[name]["spectrum_rel_err_vs_complex128_data"] = max(
                float(np.abs(o.cpu().numpy() - ref128).max() / np.abs(ref128).max()) for o in outs)
    return out


def cpu_baseline(x, t, N, lb, budget_s, nv_full):
    """The CPU oracle (numpy/scipy restatement issuing the reference's library calls, complex128 like the
    reference, ONE core) on a bounded sample of the same workload: chunks of 2048 voxels are processed until
    ~budget_s seconds of CPU work have been spent.  Per chunk: pad -> window -> ortho FFT -> roll -> |X|
    arg-max (timed) and the broadcast phase multiply (timed, with the final table's cost); the O(1)
    differential-evolution solve on the best slice is timed once.  value = projection to the full voxel
    count: nv / (nv * per_spectrum + DE)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xmris_oracle as orc

    chunk, done, t_stream = 2048, 0, 0.0
    best = (-1.0, None, None)
    info = None
    while done < x.shape[0] and t_stream < budget_s:
        xs = x[done:done + chunk].cpu().numpy().astype(np.complex128)
        t0 = time.perf_counter()
        spec, inf = orc.pipeline_values(xs, t, N, lb, solve=False)
        amax = float(np.abs(inf["slice"]).max())
        t1 = time.perf_counter()
        dummy = orc.phase_values(spec, inf["freq"], 1, 10.0, 20.0, inf["pivot"])  # same cost as the final multiply
        t2 = time.perf_counter()
        del dummy, spec
        t_stream += t2 - t0
        done += xs.shape[0]
        if amax > best[0]:
            best, info = (amax, inf["slice"], inf["target_idx"]), inf
    t0 = time.perf_counter()
    p0, p1, opt = orc.autophase_solve(info["slice"], info["freq"], info["pivot"], info["target_idx"], 1)
    t_de = time.perf_counter() - t0
    per_spec = t_stream / done
    full = nv_full * per_spec + t_de
    return {
        "value": nv_full / full, "unit": "spectra/s", "cores": 1, "kind": "port",
        "sample": f"oracle (numpy {np.__version__} pocketfft + scipy DE, complex128) on the first {done} of {nv_full} voxels "
                  f"({t_stream:.1f} s of streaming work): {per_spec * 1e3:.3f} ms/spectrum, DE solve {t_de:.3f} s once per "
                  f"dataset ({int(opt.nfev)} evals); value = {nv_full} / ({nv_full} x per-spectrum + DE)",
        "streaming_spectra_per_s": 1.0 / per_spec, "de_solve_s": t_de, "host_cpus": os.cpu_count(),
    }


def _cpu_worker(shm_name, shape, t, N, lb, tasks, results):
    """Worker process of the all-cores CPU baseline (forked before the parent initialised the GPU): the oracle's
    streaming stages on row ranges of the shared sample, one numpy thread per process like a reference user's."""
    from multiprocessing import shared_memory

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import xmris_oracle as orc

    shm = shared_memory.SharedMemory(name=shm_name)
    xs = np.ndarray(shape, dtype=np.complex64, buffer=shm.buf)
    while True:
        job = tasks.get()
        if job is None:
            break
        lo, hi = job
        t0 = time.perf_counter()
        spec, inf = orc.pipeline_values(xs[lo:hi].astype(np.complex128), t, N, lb, solve=False)
        amax = float(np.abs(inf["slice"]).max())
        orc.phase_values(spec, inf["freq"], 1, 10.0, 20.0, inf["pivot"])
        results.put((hi - lo, time.perf_counter() - t0, amax))
    shm.close()


class CpuPool:
    """the oracle with the voxel axis sharded over the host's cores by a PROCESS pool.  The
    workers are forked at the very start of the run -- before anything touches the GPU (a process that has initialised
    HIP must not fork) -- and sleep on a queue until the GPU measurements are done."""

    def __init__(self, n_sample, n_time, t, N, lb):
        from multiprocessing import get_context, shared_memory

        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            cpus = int(int(quota) / int(period)) if quota != "max" else 1 << 30
        except (OSError, ValueError):
            cpus = 1 << 30
        self.workers = max(1, min(cpus, len(os.sched_getaffinity(0)), 64))
        self.shape = (n_sample, n_time)
        self.shm = shared_memory.SharedMemory(create=True, size=n_sample * n_time * 8)
        ctx = get_context("fork")
        self.tasks, self.results = ctx.Queue(), ctx.Queue()
        self.procs = [ctx.Process(target=_cpu_worker, args=(self.shm.name, self.shape, t, N, lb, self.tasks, self.results),
                                  daemon=True) for _ in range(self.workers)]
        for p in self.procs:
            p.start()

    def run(self, sample, nv_full, t_de, chunk=256):
        """`sample`: complex64 host array of self.shape (the first rows of the benchmark's dataset)."""
        n_sample = self.shape[0]
        np.ndarray(self.shape, dtype=np.complex64, buffer=self.shm.buf)[:] = sample
        jobs = [(lo, min(n_sample, lo + chunk)) for lo in range(0, n_sample, chunk)]
        t0 = time.perf_counter()
        for j in jobs:
            self.tasks.put(j)
        done = 0
        for _ in jobs:
            done += self.results.get(timeout=600)[0]
        wall = time.perf_counter() - t0
        per_spec = wall / done
        return {
            "value": nv_full / (nv_full * per_spec + t_de), "unit": "spectra/s", "cores": self.workers, "kind": "port",
            "pool": "processes, forked before the GPU was initialised; one numpy thread each",
            "sample": f"oracle sharded over {self.workers} worker processes in chunks of {chunk} voxels, first {n_sample} of "
                      f"{nv_full} voxels in {wall:.2f} s wall; DE solve {t_de:.3f} s once per dataset (single-threaded scipy)",
            "streaming_spectra_per_s": 1.0 / per_spec,
        }

    def close(self):
        if self.procs is None:
            return
        procs, self.procs = self.procs, None
        for _ in procs:
            self.tasks.put(None)
        for p in procs:
            p.join(timeout=10)
        try:
            self.shm.close()
            self.shm.unlink()
        except OSError:
            pass


if __name__ == "__main__":
    main()
