"""Row-sharded gallery search over torch.distributed (RCCL on GPUs, gloo in CPU tests).

The path shards two ways: crops are split data-parallel (no collective: every rank
embeds its own batch), and the gallery is split by rows (rank r holds global rows
[row_base_r, row_base_r + rows_r)).  The two meet in one exchange per step:

  1. all-gather the per-rank unit fp16 queries [B, D] -> [P*B, D];
  2. every rank scans ITS shard with all P*B queries -> per-shard top-k with global rows;
  3. all-gather the per-shard lists [P*B, k] -> [P, P*B, k] (rows and scores in one buffer);
  4. merge the P lists of every query -> global top-k (same tie rule as one shard:
     descending score, ascending row).

The scan and the merge are callables so that the same orchestration runs on the
HIP library (bench.py) and on a CPU stand-in (tests, gloo).
"""
from __future__ import annotations

from typing import Callable

import torch
import torch.distributed as dist


class ShardedSearch:
    def __init__(self, world: int, batch: int, dim: int, k: int, device,
                 search_local: Callable, merge: Callable, qdtype=torch.float16, always_exchange: bool = False,
                 drain_before_exchange: bool = False, host_staged: bool = False):
        """search_local(q_all, scores_out, rows_out[, qkeys_all]): fill [P*B,k] float32 / int64 (global
        rows, -1 = none); the fourth argument (int32 [P*B], same device) is passed only by calls that
        carry keys.  merge(s_all, r_all, out_s, out_r): [P,P*B,k] -> [P*B,k]; s_all and r_all are
        strided views into the gathered blocks (rank p's block starts `block_bytes` after rank
        p-1's, rows first, then scores: r_all.data_ptr() is the address of the whole buffer), the
        layout frhot_merge_topk_packed_dev reads."""
        self.P, self.B, self.D, self.k = world, batch, dim, k
        self.search_local, self.merge = search_local, merge
        # always_exchange: run the gathers and the merge even with one rank (a rehearsal of the
        # multi-rank code path on a single GPU)
        self.exchange = world > 1 or always_exchange
        # drain_before_exchange: wait on the host for the device before every collective.  Only for
        # rehearsals in which several ranks share ONE GPU over gloo: there a gather queued behind
        # pending kernels was seen to stall for minutes (the processes' queues are time-sliced),
        # while a gather on an idle device completes at once.  One rank per GPU never needs it.
        self.drain = drain_before_exchange
        # host_staged: the collectives run on host copies of the device buffers (a backend without
        # device support: gloo, whose own staging of device tensors was seen to stall now and then
        # between processes that share one GPU).  RCCL gathers the device buffers themselves.
        self.host_staged = host_staged and torch.device(device).type != "cpu"
        n = world * batch
        self.q_all = torch.empty((n, dim), dtype=qdtype, device=device) if self.exchange else None
        # A shard's answer travels as ONE block: its rows (int64 [n,k]) followed by its scores
        # (float32 [n,k]), so a step makes two collectives (queries out, answers back), both
        # latency-bound.  The block is allocated as int64 words (an odd n*k leaves four bytes of
        # padding at its end), which keeps every view below aligned whatever n and k are.
        nk = n * k
        self._words = nk + (nk + 1) // 2
        self.block_bytes = self._words * 8
        self.packed = torch.empty(self._words, dtype=torch.int64, device=device)
        self.r_local = self.packed[:nk].view(n, k)
        self.s_local = self.packed[nk:].view(torch.float32)[:nk].view(n, k)
        if self.exchange:
            # the gathered blocks, rank-major; the merge reads rows and scores in place through
            # strided views (stride(0) = one block), no unpacking copies
            self.packed_all = torch.empty((world, self._words), dtype=torch.int64, device=device)
            self.r_all = self.packed_all[:, :nk].view(world, n, k)
            self.s_all = self.packed_all[:, nk:].view(torch.float32)[:, :nk].view(world, n, k)
            self.out_s = torch.empty((n, k), dtype=torch.float32, device=device)
            self.out_r = torch.empty((n, k), dtype=torch.int64, device=device)
            self.k_all = torch.empty(n, dtype=torch.int32, device=device)
            if self.host_staged:
                self._q_h = torch.empty((n, dim), dtype=qdtype)
                self._k_h = torch.empty(n, dtype=torch.int32)
                self._p_h = torch.empty((world, self._words), dtype=torch.int64)

    def __call__(self, q_local: torch.Tensor, qkeys_local: torch.Tensor | None = None):
        """q_local [B, D] unit rows of this rank (and, for a search inside collections, their int32
        keys [B] on the same device; every rank then passes keys) -> (scores [P*B,k], rows [P*B,k])
        for every rank's queries.  Keys cost one more small all-gather."""
        if not self.exchange:
            if qkeys_local is None:
                self.search_local(q_local, self.s_local, self.r_local)
            else:
                self.search_local(q_local, self.s_local, self.r_local, qkeys_local)
            return self.s_local, self.r_local
        self._drain()
        self._gather(self.q_all, q_local, "_q_h")
        if qkeys_local is None:
            self.search_local(self.q_all, self.s_local, self.r_local)
        else:
            self._drain()
            self._gather(self.k_all, qkeys_local.to(torch.int32).contiguous(), "_k_h")
            self.search_local(self.q_all, self.s_local, self.r_local, self.k_all)
        self._drain()
        # the gather output is passed flat (rank-major): gloo rejects the stacked shape
        self._gather(self.packed_all.view(-1), self.packed, "_p_h")
        self.merge(self.s_all, self.r_all, self.out_s, self.out_r)
        return self.out_s, self.out_r


    def _gather(self, out: torch.Tensor, mine: torch.Tensor, mirror: str):
        if not self.host_staged:
            dist.all_gather_into_tensor(out, mine)
            return
        h = getattr(self, mirror)
        dist.all_gather_into_tensor(h.view(-1), mine.cpu().view(-1))      # (.cpu() waits for the producer)
        out.view(-1).copy_(h.view(-1))

    def _drain(self):
        # everything this process has queued on the device, on any stream (the staging copies of the
        # collective before this one included)
        if self.drain and self.q_all.is_cuda:
            torch.cuda.synchronize(self.q_all.device)


def shard_rows(total_rows: int, world: int, rank: int):
    """Contiguous row split: (row_base, rows_local); the last ranks may hold one row fewer."""
    per = (total_rows + world - 1) // world
    base = min(rank * per, total_rows)
    return base, max(0, min(per, total_rows - base))
