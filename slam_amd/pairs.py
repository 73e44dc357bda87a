"""Sharding of keyframe-pair matching over ranks and the gather of its results.

Pairs are independent, so they are dealt to ranks in equal contiguous ranges of
the pair list and computed with no data-path collective.  The only exchange is
the gather of fixed-size result rows (one 32-bit word per query descriptor:
distance << 16 | train index, 0xFFFFFFFF = no match) to rank 0, issued chunk by
chunk on a second stream so that it overlaps the next chunk's compute.  Works on
any torch.distributed backend ("nccl" = RCCL on the GPUs, "gloo" in CPU tests).
"""
import numpy as np


def all_pairs(n_keyframes):
    """Every (i, j) with i < j, row-major, as an int32 array of shape [P, 2]."""
    iu = np.triu_indices(n_keyframes, 1)
    return np.stack([iu[0], iu[1]], axis=1).astype(np.int32)


def shard(n_pairs, world, rank):
    """Contiguous share of the pair list: (lo, hi, rows_per_rank)."""
    per = (n_pairs + world - 1) // world
    lo = min(rank * per, n_pairs)
    hi = min((rank + 1) * per, n_pairs)
    return lo, hi, per


def chunk_rows(per, chunks, c):
    """Row range of chunk c inside a rank's (padded) result buffer; equal on every rank."""
    return per * c // chunks, per * (c + 1) // chunks


class PairGather:
    """Per-rank result buffer [per, n_per] int32 plus the chunked gather to rank 0."""

    def __init__(self, torch, dist, n_pairs, n_per, world, rank, device, chunks, collective=None, group=None):
        """collective: run the gather even for world == 1 (a one-rank process group, to
        exercise the exchange on a single GPU); default: only when world > 1.
        group: the process group that carries the gather (default: the default group)."""
        self.torch, self.dist, self.group = torch, dist, group
        self.world, self.rank, self.chunks = world, rank, max(1, chunks)
        self.collective = (world > 1) if collective is None else bool(collective)
        self.lo, self.hi, self.per = shard(n_pairs, world, rank)
        self.mine = self.hi - self.lo
        self.n_pairs, self.n_per = n_pairs, n_per
        self.res = torch.full((max(self.per, 1), n_per), -1, dtype=torch.int32, device=device)
        self.gathered = None
        if self.collective and rank == 0:
            self.gathered = torch.empty((world, max(self.per, 1), n_per), dtype=torch.int32, device=device)
        self.on_gpu = (getattr(device, "type", str(device)) == "cuda")
        self.comm = torch.cuda.Stream(device=device) if (self.on_gpu and self.collective) else None

    def run(self, compute, limit=None):
        """compute(a, b) fills rows [a, b) of self.res (local pair indices lo+a .. lo+b).

        limit: compute at most that many rows per chunk and skip the gather (warm-up).
        """
        torch, dist = self.torch, self.dist
        handles = []
        for c in range(self.chunks):
            ca, cb = chunk_rows(self.per, self.chunks, c)
            a, b = min(ca, self.mine), min(cb, self.mine)
            if limit is not None:
                b = min(b, a + limit)
            if b > a:
                compute(a, b)
            if self.collective and limit is None and cb > ca:
                src = self.res[ca:cb]
                outs = [self.gathered[r, ca:cb] for r in range(self.world)] if self.rank == 0 else None
                if self.comm is not None:
                    ev = torch.cuda.Event()
                    ev.record()
                    with torch.cuda.stream(self.comm):
                        self.comm.wait_event(ev)
                        handles.append(dist.gather(src, outs, dst=0, group=self.group, async_op=True))
                else:
                    handles.append(dist.gather(src, outs, dst=0, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        if self.comm is not None:
            torch.cuda.current_stream().wait_stream(self.comm)

    def result(self):
        """On rank 0: [n_pairs, n_per] rows in pair-list order (a view when world == 1)."""
        if not self.collective:
            return self.res[:self.n_pairs]
        if self.rank != 0:
            return None
        return self.gathered.reshape(self.world * self.per, self.n_per)[:self.n_pairs]
