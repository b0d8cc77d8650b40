"""Row-sharded search across the GPUs of one node: one process per GPU.

The reference is single-process (nothing in it shards), so this module has no
counterpart there; it keeps the single-GPU contract: the merged result equals
what one GPU holding all rows would return, including the tie rule (equal
distances go to the lower global row number).

Partition: rank r owns the contiguous global rows [r*N//G, (r+1)*N//G).  Every
query goes to every rank; each rank returns its local k best with GLOBAL row
numbers; one all-gather (RCCL over xGMI when the tensors are on GPUs) moves
G*nq*k (distance, row) pairs, and each rank merges them with the same
selection kernel that merges the per-block lists of a single-GPU search.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Global row range [lo, hi) owned by `rank`."""
    return (rank * n_total) // world, ((rank + 1) * n_total) // world


class HipFlatBackend:
    """Local search and merge on the GPU through the C ABI."""

    def __init__(self, dim: int, metric: int):
        from . import _native
        self._n = _native
        self.store = _native.RowStore(dim)
        self.metric = metric

    def add_device(self, rows: torch.Tensor) -> None:
        assert rows.is_cuda and rows.dtype == torch.float32 and rows.is_contiguous()
        self.store.append_device(rows.data_ptr(), rows.shape[0], rows.shape[1],
                                 torch.cuda.current_stream().cuda_stream)

    def local_search(self, queries: torch.Tensor, k: int, out_dist: torch.Tensor, out_rows: torch.Tensor) -> None:
        self.store.search_device(queries.data_ptr(), queries.shape[0], k, self.metric,
                                 out_dist.data_ptr(), out_rows.data_ptr(),
                                 stream=torch.cuda.current_stream().cuda_stream)

    def merge(self, all_dist: torch.Tensor, all_rows: torch.Tensor, k: int,
              out_dist: torch.Tensor, out_rows: torch.Tensor) -> None:
        world, nq = all_dist.shape[0], all_dist.shape[1]
        self._n.topk_merge_device(all_dist.data_ptr(), all_rows.data_ptr(), world, nq, k,
                                  out_dist.data_ptr(), out_rows.data_ptr(),
                                  stream=torch.cuda.current_stream().cuda_stream)


class ShardedFlat:
    """Flat search over rows sharded across the ranks of a process group."""

    def __init__(self, backend, n_total: int, rank: int = 0, world: int = 1, group=None):
        self.backend = backend
        self.rank, self.world, self.group = rank, world, group
        self.lo, self.hi = shard_range(n_total, world, rank)
        self._buf = {}

    def _buffers(self, nq: int, k: int, device):
        key = (nq, k, str(device))
        if key not in self._buf:
            f = dict(device=device, dtype=torch.float32)
            i = dict(device=device, dtype=torch.int64)
            self._buf[key] = (torch.empty((nq, k), **f), torch.empty((nq, k), **i),
                              torch.empty((self.world, nq, k), **f), torch.empty((self.world, nq, k), **i),
                              torch.empty((nq, k), **f), torch.empty((nq, k), **i))
        return self._buf[key]

    def search(self, queries: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        nq = queries.shape[0]
        ld, lr, gd, gr, od, orow = self._buffers(nq, k, queries.device)
        self.backend.local_search(queries, k, ld, lr)
        if self.lo:
            lr.add_((lr >= 0).to(torch.int64) * self.lo)       # local -> global row numbers; -1 stays -1
        if self.world == 1:
            return ld, lr
        # output = the ranks' (nq, k) blocks concatenated along dim 0, i.e. [rank][query][k]
        dist.all_gather_into_tensor(gd.view(self.world * nq, k), ld, group=self.group)
        dist.all_gather_into_tensor(gr.view(self.world * nq, k), lr, group=self.group)
        self.backend.merge(gd, gr, k, od, orow)
        return od, orow
