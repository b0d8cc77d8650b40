"""Sharded search across the GPUs of one node: one process per GPU.

The reference is single-process (nothing in it shards), so this module has no
counterpart there; it keeps the single-GPU contract: the merged result equals
what one GPU holding everything would return, including the tie rule (Flat and
PQ: equal distances go to the lower global row number; IVF: to the candidate
the reference's scan visits first).

Flat rows and PQ codes shard in contiguous row blocks (ShardedFlat with a
HipFlatBackend or HipPqBackend).  IVF shards by whole inverted lists, balanced
greedily on list length, with the centroids replicated (ShardedIVF).  K-means
training all-reduces the per-cluster sums and counts (kmeans_update_allreduce).

Partition: rank r owns the contiguous global rows [r*N//G, (r+1)*N//G).  Every
query goes to every rank; each rank returns its local k best with LOCAL row
numbers; one all-gather (RCCL over xGMI when the tensors are on GPUs) moves
G*nq*k (distance, row) pairs, and each rank merges them by (distance, global
row), adding every rank's first global row on the way.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Global row range [lo, hi) owned by `rank`."""
    return (rank * n_total) // world, ((rank + 1) * n_total) // world


def _mask_ptr(mask_words: Optional[torch.Tensor], n_rows: int) -> int:
    if mask_words is None:
        return 0
    if not (mask_words.is_cuda and mask_words.is_contiguous() and mask_words.element_size() == 4
            and mask_words.numel() * 32 >= n_rows):
        raise ValueError(f"a filter over {n_rows} local rows takes {(n_rows + 31) // 32} contiguous 32-bit words on the GPU")
    return mask_words.data_ptr()


class HipFlatBackend:
    """Local search and merge on the GPU through the C ABI."""

    def __init__(self, dim: int, metric: int, storage: str = "f32", prefilter: Optional[str] = "auto"):
        from . import _native
        self._n = _native
        self.store = _native.RowStore(dim, storage=storage)
        if prefilter is not None and storage == "f32":
            # same results, the pass over all rows on the bf16 matrix cores ("auto": while the bf16 copy leaves room)
            self.store.set_prefilter(prefilter)
        self.metric = metric

    def add_device(self, rows: torch.Tensor) -> None:
        assert rows.is_cuda and rows.dtype == torch.float32 and rows.is_contiguous()
        self.store.append_device(rows.data_ptr(), rows.shape[0], rows.shape[1],
                                 torch.cuda.current_stream().cuda_stream)

    def local_search(self, queries: torch.Tensor, k: int, out_dist: torch.Tensor, out_rows: torch.Tensor,
                     mask_words: Optional[torch.Tensor] = None) -> None:
        """mask_words: packed filter bits over this rank's LOCAL rows (bit i of word i // 32), resident on the GPU."""
        self.store.search_device(queries.data_ptr(), queries.shape[0], k, self.metric,
                                 out_dist.data_ptr(), out_rows.data_ptr(),
                                 mask_ptr=_mask_ptr(mask_words, len(self.store)),
                                 stream=torch.cuda.current_stream().cuda_stream)

    def merge(self, all_dist: torch.Tensor, all_rows: torch.Tensor, k: int,
              out_dist: torch.Tensor, out_rows: torch.Tensor, row_base: Optional[torch.Tensor] = None) -> None:
        """all_dist / all_rows: [world][nq][k] views whose rank stride may exceed nq*k (blocks of a packed
        gather); row_base[r] is added to the rows of rank r."""
        world, nq = all_dist.shape[0], all_dist.shape[1]
        assert all_dist.stride(1) == k and all_dist.stride(2) == 1 and all_rows.stride(1) == k and all_rows.stride(2) == 1
        self._n.topk_merge_strided_device(all_dist.data_ptr(), all_rows.data_ptr(),
                                          all_dist.stride(0) if world > 1 else nq * k,
                                          all_rows.stride(0) if world > 1 else nq * k,
                                          row_base.data_ptr() if row_base is not None else 0, world, nq, k,
                                          out_dist.data_ptr(), out_rows.data_ptr(),
                                          stream=torch.cuda.current_stream().cuda_stream)


class HipPqBackend:
    """Local ADC search over this rank's block of PQ codes, merge on the GPU."""

    def __init__(self, dim: int, m: int, ksub: int, codebook):
        from . import _native
        self._n = _native
        self.codes = _native.PqCodes(dim, m, ksub)
        self.codes.set_codebook(codebook)

    def add_device(self, codes: torch.Tensor) -> None:
        assert codes.is_cuda and codes.dtype == torch.uint8 and codes.is_contiguous()
        self.codes.append_codes_device(codes.data_ptr(), codes.shape[0], torch.cuda.current_stream().cuda_stream)

    def local_search(self, queries: torch.Tensor, k: int, out_dist: torch.Tensor, out_rows: torch.Tensor,
                     mask_words: Optional[torch.Tensor] = None) -> None:
        self.codes.search_device(queries.data_ptr(), queries.shape[0], k, out_dist.data_ptr(), out_rows.data_ptr(),
                                 mask_ptr=_mask_ptr(mask_words, len(self.codes)),
                                 stream=torch.cuda.current_stream().cuda_stream)

    merge = HipFlatBackend.merge


class ShardedFlat:
    """Flat search over rows sharded across the ranks of a process group.

    A rank's result travels as one packed block, [nq*k fp32 distances | pad to 8 bytes | nq*k int64
    LOCAL rows], so one all-gather moves everything; the merge reads the gathered blocks in place
    and adds every rank's first global row to its rows."""

    def __init__(self, backend, n_total: int, rank: int = 0, world: int = 1, group=None):
        self.backend = backend
        self.rank, self.world, self.group = rank, world, group
        self.n_total = n_total
        self.lo, self.hi = shard_range(n_total, world, rank)
        self._buf = {}

    def _buffers(self, nq: int, k: int, device):
        key = (nq, k, str(device))
        if key not in self._buf:
            dist_bytes = (nq * k * 4 + 7) // 8 * 8
            block = dist_bytes + nq * k * 8
            local = torch.empty(block, device=device, dtype=torch.uint8)
            gathered = torch.empty(self.world * block, device=device, dtype=torch.uint8)
            ld = local[:nq * k * 4].view(torch.float32).view(nq, k)
            lr = local[dist_bytes:].view(torch.int64).view(nq, k)
            g2 = gathered.view(self.world, block)
            gd = g2[:, :nq * k * 4].view(torch.float32).view(self.world, nq, k)      # rank stride = block / 4 floats
            gr = g2[:, dist_bytes:].view(torch.int64).view(self.world, nq, k)       # rank stride = block / 8 rows
            base = torch.tensor([shard_range(self.n_total, self.world, r)[0] for r in range(self.world)],
                                device=device, dtype=torch.int64)
            self._buf[key] = (local, gathered, ld, lr, gd, gr, base,
                              torch.empty((nq, k), device=device, dtype=torch.float32),
                              torch.empty((nq, k), device=device, dtype=torch.int64))
        return self._buf[key]

    def search(self, queries: torch.Tensor, k: int, mask_words: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """mask_words: this rank's slice of the filter, packed over its local rows (see the backends)."""
        nq = queries.shape[0]
        local, gathered, ld, lr, gd, gr, base, od, orow = self._buffers(nq, k, queries.device)
        if mask_words is None:
            self.backend.local_search(queries, k, ld, lr)
        else:
            self.backend.local_search(queries, k, ld, lr, mask_words)
        if self.world == 1:
            return ld, lr
        dist.all_gather_into_tensor(gathered, local, group=self.group)
        self.backend.merge(gd, gr, k, od, orow, base)
        return od, orow


# --------------------------------------------------------------------------- IVF
def assign_lists_balanced(list_lengths, world: int):
    """Owner rank of every inverted list: longest list first, each to the rank with
    the fewest rows so far (ties to the lower rank).  Deterministic, so every rank
    computes the same map."""
    import numpy as np
    lengths = np.asarray(list_lengths, dtype=np.int64)
    owner = np.empty(len(lengths), dtype=np.int32)
    load = np.zeros(world, dtype=np.int64)
    for l in np.argsort(-lengths, kind="stable"):
        r = int(np.argmin(load))
        owner[l] = r
        load[r] += lengths[l]
    return owner


def local_ivf_layout(list_of_row, n_lists: int, owner, rank: int):
    """For this rank: the global rows it stores (ascending), and the CSR layout of its
    lists over those rows (`list_rows` in local numbering, `list_offsets`); lists of
    other ranks are empty.  Positions inside a list are the same as on a single GPU."""
    import numpy as np
    list_of_row = np.asarray(list_of_row)
    mine = np.flatnonzero(owner[list_of_row] == rank)                  # global rows, ascending
    order = mine[np.argsort(list_of_row[mine], kind="stable")]        # grouped by list, insertion order inside
    counts = np.bincount(list_of_row[mine], minlength=n_lists).astype(np.int64)
    offsets = np.zeros(n_lists + 1, dtype=np.int64)
    np.cumsum(counts, out=offsets[1:])
    local_of_global = np.full(len(list_of_row), -1, dtype=np.int64)
    local_of_global[mine] = np.arange(len(mine))
    return mine, local_of_global[order], offsets


class HipIvfBackend:
    """This rank's inverted lists behind the C ABI (all centroids, own lists only).  The rows are stored under
    their GLOBAL row numbers, so a local search returns global rows and nothing is translated afterwards."""

    def __init__(self, dim: int, metric: int, centroids, rows_local, list_rows_local, list_offsets, global_rows):
        import numpy as np
        from . import _native
        self._n = _native
        self.lists = _native.IvfLists(dim, metric)
        self.lists.set_centroids(centroids)
        order = np.asarray(list_rows_local, dtype=np.int64)            # local rows in list order
        rows_local = np.ascontiguousarray(rows_local, dtype=np.float32).reshape(-1, dim)
        self.lists.build_numbered(rows_local[order], np.asarray(global_rows, dtype=np.int64)[order], list_offsets)

    def local_search(self, queries, k, n_probe, out_dist, out_rows, out_order) -> None:
        self.lists.search_device(queries.data_ptr(), queries.shape[0], k, n_probe, out_dist.data_ptr(), out_rows.data_ptr(),
                                 stream=torch.cuda.current_stream().cuda_stream, out_order_ptr=out_order.data_ptr())

    def merge(self, all_dist, all_order, all_rows, k, out_dist, out_rows) -> None:
        """all_* : [world][nq][k] views into the gathered blocks (rank stride = one block)."""
        world, nq = all_dist.shape[0], all_dist.shape[1]
        assert all_dist.stride(1) == k and all_order.stride(1) == k and all_rows.stride(1) == k
        assert all_order.stride(0) == all_rows.stride(0) or world == 1
        self._n.topk_merge_keyed_strided_device(all_dist.data_ptr(), all_order.data_ptr(), all_rows.data_ptr(),
                                                all_dist.stride(0) if world > 1 else nq * k,
                                                all_order.stride(0) if world > 1 else nq * k, world, nq, k,
                                                out_dist.data_ptr(), out_rows.data_ptr(),
                                                stream=torch.cuda.current_stream().cuda_stream)


class ShardedIVF:
    """IVF search with the inverted lists spread over the ranks of a process group.
    Every rank ranks all centroids (they are replicated), scans the probed lists it
    owns, and the per-rank results are merged by (distance, visiting order).

    A rank's result travels as one packed block, [nq*k fp32 distances | pad to 8 bytes | nq*k int64 visiting
    orders | nq*k int64 global rows]: one all-gather per search, merged where the blocks arrive.  The buffers
    are made once per (nq, k, device) and reused."""

    def __init__(self, backend, rank: int = 0, world: int = 1, group=None):
        self.backend, self.rank, self.world, self.group = backend, rank, world, group
        self._buf = {}

    def _buffers(self, nq: int, k: int, device):
        key = (nq, k, str(device))
        if key not in self._buf:
            dist_bytes = (nq * k * 4 + 7) // 8 * 8
            block = dist_bytes + 2 * nq * k * 8
            local = torch.empty(block, device=device, dtype=torch.uint8)
            gathered = torch.empty(self.world * block, device=device, dtype=torch.uint8)

            def views(buf, lead):
                b2 = buf.view(lead, block)
                return (b2[:, :nq * k * 4].view(torch.float32).view(lead, nq, k),
                        b2[:, dist_bytes:dist_bytes + nq * k * 8].view(torch.int64).view(lead, nq, k),
                        b2[:, dist_bytes + nq * k * 8:].view(torch.int64).view(lead, nq, k))
            ld, lo, lr = (v[0] for v in views(local, 1))
            gd, go, gr = views(gathered, self.world)
            self._buf[key] = (local, gathered, ld, lo, lr, gd, go, gr,
                              torch.empty((nq, k), device=device, dtype=torch.float32),
                              torch.empty((nq, k), device=device, dtype=torch.int64))
        return self._buf[key]

    def search(self, queries: torch.Tensor, k: int, n_probe: int):
        nq = queries.shape[0]
        local, gathered, ld, lo, lr, gd, go, gr, od, orow = self._buffers(nq, k, queries.device)
        self.backend.local_search(queries, k, n_probe, ld, lr, lo)
        if self.world == 1:
            return ld, lr
        dist.all_gather_into_tensor(gathered, local, group=self.group)
        self.backend.merge(gd, go, gr, k, od, orow)
        return od, orow


# --------------------------------------------------------------------------- k-means
def kmeans_update_allreduce(sums: torch.Tensor, counts: torch.Tensor, old_centroids: torch.Tensor, group=None):
    """One Lloyd update over sharded training rows: all-reduce the per-cluster fp64
    sums [k][d] and counts [k] that every rank accumulated over its own rows, then
    mean; a cluster nobody feeds keeps its centroid (the caller reseeds it the way
    index/ivf.py:192-196 does).  Returns (centroids fp32, global counts)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sums, group=group)
        dist.all_reduce(counts, group=group)
    fed = counts > 0
    new = old_centroids.clone()
    new[fed] = (sums[fed] / counts[fed].unsqueeze(1).to(sums.dtype)).to(old_centroids.dtype)
    return new, counts
