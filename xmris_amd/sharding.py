"""Voxel-axis sharding over the GPUs of one node (one process per GPU, ``torch.distributed``).

The hot path is independent per spectrum except for autophase's GLOBAL arg-max and the single
(p0, p1, pivot) it yields (reference ``phasing.py:229, 276-290``).  So the only cross-rank traffic is
O(1) per dataset: one all_gather of (max |X|, global flat index) per rank and one broadcast of
(p0, p1) from the rank that owns the winning spectrum.  No data-path collective exists.
"""
from __future__ import annotations


def shard_bounds(n_rows: int, world: int, rank: int):
    """Contiguous block of the flattened non-FID axes owned by `rank`: rows [lo, hi)."""
    lo = (n_rows * rank) // world
    hi = (n_rows * (rank + 1)) // world
    return lo, hi


def pick_winner(pairs):
    """pairs[r] = (max_abs, global_flat_index) of rank r.  np.argmax semantics over the whole
    dataset: the largest magnitude wins, ties go to the lowest global flat index.  Returns
    (owner_rank, global_flat_index, max_abs)."""
    best_r, best = 0, pairs[0]
    for r, p in enumerate(pairs):
        if p[0] > best[0] or (p[0] == best[0] and p[1] < best[1]):
            best_r, best = r, p
    return best_r, int(best[1]), float(best[0])


def exchange_argmax(max_abs: float, global_flat: int, dist=None, device="cpu", group=None):
    """All ranks learn the winner.  `dist` = an initialised torch.distributed module (or None for a
    single process).  Works on RCCL ("nccl", GPU tensors) and gloo (CPU tensors, `group` = a gloo
    process group).  For a streaming pipeline prefer a gloo group: an RCCL collective is ordered behind
    everything already queued on the GPU (other datasets' kernels), a 32-byte host exchange is not."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0, int(global_flat), float(max_abs)
    import torch

    world = dist.get_world_size()
    # the flat index needs 64 integer bits (2^19 voxels x 2^13 points x 8 GPUs > 2^32): ship it as int64
    # and the magnitude as the bit pattern of a float64
    mine = torch.tensor([torch.tensor(max_abs, dtype=torch.float64).view(torch.int64).item(), int(global_flat)],
                        dtype=torch.int64, device=device)
    allv = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    pairs = []
    for a in allv:
        a = a.cpu()
        pairs.append((float(a[0:1].view(torch.float64).item()), int(a[1].item())))
    return pick_winner(pairs)


def broadcast_params(values, owner: int, dist=None, device="cpu", group=None):
    """Broadcast a short list of float64 parameters (p0, p1) from `owner` to every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(v) for v in values]
    import torch

    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.broadcast(t, src=owner, group=group)
    return [float(v) for v in t.cpu()]


class ShmExchange:
    """The same O(1) exchange for the ranks of ONE node through a 4 KiB shared-memory page instead of
    loopback sockets: a gloo all_gather + broadcast costs ~0.5 ms per dataset, two spins on a cache line
    cost microseconds, and the exchange sits on the critical path of every dataset (the winner's search
    cannot start before it).  Created collectively (`ShmExchange.create(dist, group)`): rank 0 makes a file
    under /dev/shm, every rank maps it, rank 0 unlinks it at once -- nothing is left behind, even on a crash.

    Layout (int64 words, float64 stored as bit patterns): BANKS banks of `world` gather slots
    [seq, max_abs, flat, -] and one broadcast slot [seq, n, v0, v1, ...].  Gathers and broadcasts carry their OWN
    sequence numbers (a streaming caller runs the gathers of later datasets before the broadcast of an earlier
    one) and pick their bank by that number.  A rank can run at most one gather ahead of the slowest one (it needs
    everyone's entry to finish the next gather), and callers put at least one gather between two broadcasts, so
    two banks would do; four leave slack.  Publication order: the payload words are written with plain stores, the
    sequence word that publishes them with a RELEASE store and every poll reads it with ACQUIRE loads -- both in
    the host library (`xm_atomic_store_release_i64`, `xm_atomic_wait_all_ge_i64`), so the protocol leans neither on
    x86's store ordering nor on what the interpreter does between two numpy stores."""

    SLOT = 8   # int64 words per slot = one 64-byte cache line
    BANKS = 4

    def __init__(self, buf, rank: int, world: int, timeout_s: float = 120.0):
        import numpy as np

        self._buf = buf
        self.rank, self.world = rank, world
        self.timeout_s = timeout_s
        try:  # sleeping polls of ~20 us need a fine timer: the default 50 us timer slack would triple them
            import ctypes

            ctypes.CDLL(None, use_errno=True).prctl(29, 1000, 0, 0, 0)  # PR_SET_TIMERSLACK = 1 us, this thread
        except Exception:
            pass
        words = np.frombuffer(buf, dtype=np.int64)
        self._i = words[: self.BANKS * (world + 1) * self.SLOT].reshape(self.BANKS, world + 1, self.SLOT)
        self._f = self._i.view(np.float64)
        self._seq = 0   # gathers done
        self._bseq = 0  # broadcasts done
        self.calls = {"gather": 0, "broadcast": 0}  # per-rank call counts (tests compare them across the ranks)
        from . import _lib

        self._lib = _lib.load()
        self._base = self._i.ctypes.data

    @staticmethod
    def nbytes(world: int) -> int:
        return max(4096, ShmExchange.BANKS * (world + 1) * ShmExchange.SLOT * 8)

    @classmethod
    def create(cls, dist, group=None, timeout_s: float = 120.0):
        """Collective over `group` (a gloo group, or the default group when that is gloo): object broadcast of
        the file name + one all_reduce.  Raises OSError on EVERY rank when any rank cannot map the page."""
        import mmap
        import os
        import uuid

        import torch

        rank, world = dist.get_rank(), dist.get_world_size()
        size = cls.nbytes(world)
        name, fd, buf = [None], -1, None
        if rank == 0:
            try:
                name[0] = f"/dev/shm/xmris_amd_{os.getpid()}_{uuid.uuid4().hex}"
                fd = os.open(name[0], os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                os.ftruncate(fd, size)  # zero-filled: every sequence number starts at 0
            except OSError:
                name[0] = None
        dist.broadcast_object_list(name, src=0, group=group)
        ok = 0
        if name[0] is not None:
            try:
                if rank != 0:
                    fd = os.open(name[0], os.O_RDWR)
                buf = mmap.mmap(fd, size)
                os.close(fd)
                ok = 1
            except OSError:
                ok = 0
        # every rank learns whether EVERY rank mapped the page (this is also the "all mapped" barrier)
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if rank == 0 and name[0] is not None:
            try:
                os.unlink(name[0])
            except OSError:
                pass
        if int(flag.item()) == 0:
            raise OSError("ShmExchange: the ranks do not share a writable /dev/shm")
        return cls(buf, rank, world, timeout_s)

    def _addr(self, bank: int, slot: int, word: int = 0) -> int:
        return self._base + 8 * ((bank * (self.world + 1) + slot) * self.SLOT + word)

    def _wait(self, addr: int, count: int, value: int):
        """Until `count` sequence words (one per slot from `addr`) are >= value: a short busy phase in the library
        (ranks usually arrive together; acquire loads), then sleeping polls -- a rank that waits a millisecond for the
        owner's search must not take a core away from the search."""
        import time

        wait = self._lib.xm_atomic_wait_all_ge_i64
        if wait(addr, self.SLOT, count, value, 30) == 1:
            return
        t_end = time.monotonic() + self.timeout_s
        nap = 2e-5  # backs off to 0.15 ms: what is waited for here is a search of a millisecond or more
        while wait(addr, self.SLOT, count, value, 0) != 1:
            time.sleep(nap)
            nap = min(1.5e-4, nap * 1.5)
            if time.monotonic() > t_end:
                raise TimeoutError("ShmExchange: a rank did not arrive (is it still running?)")

    def exchange_argmax(self, max_abs: float, global_flat: int):
        """Every rank contributes (max |X|, global flat index); returns pick_winner() of all of them."""
        self._seq += 1
        s, bank = self._seq, self._seq % self.BANKS
        slots_i, slots_f = self._i[bank], self._f[bank]
        self.calls["gather"] += 1
        slots_f[self.rank, 1] = float(max_abs)
        slots_i[self.rank, 2] = int(global_flat)
        self._lib.xm_atomic_store_release_i64(self._addr(bank, self.rank), s)  # publish
        self._wait(self._addr(bank, 0), self.world, s)
        return pick_winner([(float(slots_f[r, 1]), int(slots_i[r, 2])) for r in range(self.world)])

    def broadcast_params(self, values, owner: int):
        """`owner` publishes a short list of float64 values; every rank makes the same sequence of calls."""
        self._bseq += 1
        s, bank = self._bseq, self._bseq % self.BANKS
        bi, bf = self._i[bank, self.world], self._f[bank, self.world]
        self.calls["broadcast"] += 1
        if self.rank == owner:
            vals = [float(v) for v in values]
            if len(vals) > self.SLOT - 2:
                raise ValueError("too many values for one slot")
            for j, v in enumerate(vals):
                bf[2 + j] = v
            bi[1] = len(vals)
            self._lib.xm_atomic_store_release_i64(self._addr(bank, self.world), s)  # publish
            return vals
        self._wait(self._addr(bank, self.world), 1, s)
        return [float(bf[2 + j]) for j in range(int(bi[1]))]
