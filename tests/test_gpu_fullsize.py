"""The hot path at the sizes BASELINE.json quotes, checked through properties that
do not need a full CPU run per query, plus full CPU checks on a few queries:

  * every returned distance equals the oracle's distance for that row (bit-exact on the
    pinned paths, 1e-4 relative on the dot-product paths);
  * results are sorted, row numbers are distinct and in range, filters are respected;
  * completeness: for a few queries the oracle scans everything (or the probed lists) and
    must pick the same rows;
  * the batched kernels and the single-query kernels agree;
  * running a search twice gives the same bits.

Inputs are generated on the GPU (seeded) and copied back for the oracle.
"""

import numpy as np
import pytest
import torch

from oracle import ref_numpy as R
from conftest import same_ids_or_near_ties

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(native):
    from vectordb_amd import _native
    return _native


def randn(n, d, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return torch.randn((n, d), device="cuda", dtype=torch.float32, generator=g)


def search_dev(fn, q, k):
    nq = q.shape[0]
    od = torch.empty((nq, k), device="cuda", dtype=torch.float32)
    orow = torch.empty((nq, k), device="cuda", dtype=torch.int64)
    fn(q.data_ptr(), nq, k, od.data_ptr(), orow.data_ptr())
    torch.cuda.synchronize()
    return od.cpu().numpy(), orow.cpu().numpy()


def topk_big(d, k):
    """The oracle's stable top-k (ties to the lower row) without sorting all of a 100M-entry array."""
    thr = np.partition(d, k - 1)[k - 1]
    idx = np.flatnonzero(d <= thr)
    return idx[np.lexsort((idx, d[idx]))][:k]


def check_shape_properties(dist, rows, n):
    assert np.all(np.diff(dist, axis=1) >= 0), "distances must ascend"
    assert rows.min() >= 0 and rows.max() < n
    for r in rows:
        assert len(set(r.tolist())) == len(r), "row numbers must be distinct"


# ---------------------------------------------------------------- Flat cosine 1M x 384, batch 1024 (configs[1])
def test_flat_cosine_1m_batch_1024(nat):
    n, d, nq, k = 1_000_000, 384, 1024, 10
    x = randn(n, d, 42)
    q = randn(nq, d, 123)
    store = nat.RowStore(d)
    store.append_device(x.data_ptr(), n, d, 0)
    fn = lambda qp, m, kk, dp, rp: store.search_device(qp, m, kk, nat.COSINE, dp, rp)
    dist, rows = search_dev(fn, q, k)
    dist2, rows2 = search_dev(fn, q, k)
    assert np.array_equal(rows, rows2) and np.array_equal(dist, dist2), "same call, same bits"
    check_shape_properties(dist, rows, n)

    xh, qh = x.cpu().numpy(), q.cpu().numpy()
    # every returned distance against the oracle's value for that row (1e-4 relative, BLAS vs MFMA order)
    for qi in range(0, nq, 37):
        ref = R.query_distances(qh[qi], xh[rows[qi]], "cosine")
        np.testing.assert_allclose(dist[qi], ref, rtol=1e-4, atol=2e-6)
    # completeness: full oracle scan for a few queries
    for qi in (0, 511, 1023):
        ref_all = R.query_distances(qh[qi], xh, "cosine")
        ref_rows = R.topk_stable(ref_all, k)
        assert same_ids_or_near_ties(rows[qi], ref_rows, lambda r: ref_all[r], float(ref_all[ref_rows[-1]]),
                                     rtol=1e-4, atol=2e-6)
    # the single-query streaming kernel agrees with the batched matrix-core kernel
    for qi in (3, 700):
        d1, r1 = search_dev(fn, q[qi:qi + 1].contiguous(), k)
        ref_all = R.query_distances(qh[qi], xh, "cosine")
        assert same_ids_or_near_ties(r1[0], rows[qi], lambda r: ref_all[r], float(np.sort(ref_all)[k - 1]),
                                     rtol=1e-4, atol=2e-6)
        np.testing.assert_allclose(d1[0], dist[qi], rtol=1e-4, atol=2e-6)
    store.close()


# ---------------------------------------------------------------- Manhattan 1M x 128 (configs[4] side check), bit-exact
def test_flat_manhattan_1m(nat):
    n, d, k = 1_000_000, 128, 10
    x = randn(n, d, 7)
    q = randn(9, d, 8)
    store = nat.RowStore(d)
    store.append_device(x.data_ptr(), n, d, 0)
    fn = lambda qp, m, kk, dp, rp: store.search_device(qp, m, kk, nat.MANHATTAN, dp, rp)
    dist, rows = search_dev(fn, q, k)                     # 9 queries: the 8-query kernel with a ragged second group
    check_shape_properties(dist, rows, n)
    xh, qh = x.cpu().numpy(), q.cpu().numpy()
    for qi in range(9):
        ref_all = R.query_distances(qh[qi], xh, "manhattan")
        ref_rows = R.topk_stable(ref_all, k)
        assert np.array_equal(rows[qi], ref_rows)
        assert np.array_equal(dist[qi], ref_all[ref_rows])
    d1, r1 = search_dev(fn, q[4:5].contiguous(), k)       # one query: the single-query kernel
    assert np.array_equal(r1[0], rows[4]) and np.array_equal(d1[0], dist[4])
    store.close()


# ---------------------------------------------------------------- IVF nlist=4096 nprobe=32 over 10M x 128 (configs[3])
def test_ivf_10m_nprobe_32(nat):
    n, d, nlist, nprobe, k, nq = 10_000_000, 128, 4096, 32, 10, 256
    x = randn(n, d, 11)
    xh = x.cpu().numpy()
    del x
    torch.cuda.empty_cache()
    rng = np.random.default_rng(5)
    cent = xh[rng.choice(n, nlist, replace=False)].copy()
    assign, _ = nat.assign(xh, cent, nat.EUCLIDEAN)
    # the same assignment from the oracle on a sample (identity-form batch assignment, ivf.py:352-404)
    sample = rng.choice(n, 2000, replace=False)
    assert np.array_equal(assign[sample], R.ivf_assign_batch(xh[sample], cent, "euclidean"))
    lists = nat.IvfLists(d, nat.EUCLIDEAN)
    lists.set_centroids(cent)
    lists.build(xh, assign)
    order = np.argsort(assign, kind="stable")
    bounds = np.searchsorted(assign[order], np.arange(nlist + 1))
    members = [order[bounds[l]:bounds[l + 1]] for l in range(nlist)]

    q = randn(nq, d, 12)
    qh = q.cpu().numpy()
    fn = lambda qp, m, kk, dp, rp: lists.search_device(qp, m, kk, nprobe, dp, rp)
    dist, rows = search_dev(fn, q, k)
    check_shape_properties(dist, rows, n)
    # returned distances: bit-exact direct form
    for qi in range(0, nq, 5):
        ref = R.rows_scalar_metric(qh[qi], xh[rows[qi]], "euclidean").astype(np.float32)
        assert np.array_equal(dist[qi], ref)
    # completeness and probe order: the oracle's search over the same lists
    for qi in (0, 100, 255):
        ref_rows, ref_d = R.ivf_search(qh[qi], xh, cent, members, k, nprobe, "euclidean")
        assert np.array_equal(rows[qi], ref_rows)
        assert np.array_equal(dist[qi], ref_d.astype(np.float32))
    # single-query path == batched list-major path
    for qi in (7, 200):
        d1, r1 = search_dev(fn, q[qi:qi + 1].contiguous(), k)
        assert np.array_equal(r1[0], rows[qi]) and np.array_equal(d1[0], dist[qi])
    # a filter that keeps 1 row in 16
    keep = (np.arange(n) % 16) == 3
    words = nat.pack_mask(keep)
    dm, rm = lists.search(qh[:4], k, nprobe, mask_words=words)
    assert np.all(keep[rm[rm >= 0]])
    ref_rows, ref_d = R.ivf_search(qh[1], xh, cent, members, k, nprobe, "euclidean", mask=keep)
    assert np.array_equal(rm[1][:len(ref_rows)], ref_rows)
    lists.close()


# ---------------------------------------------------------------- PQ m=16 nbits=8 over 100M codes with a filter (configs[4])
def test_pq_100m_codes_with_filter(nat):
    n, m, d, k = 100_000_000, 16, 128, 10
    rng = np.random.default_rng(3)
    book = rng.standard_normal((m, 256, d // m), dtype=np.float32)
    pq = nat.PqCodes(d, m, 256)
    pq.set_codebook(book)
    g = torch.Generator(device="cuda")
    g.manual_seed(99)
    chunk = 10_000_000
    host_codes = np.empty((n, m), np.uint8)
    for s in range(0, n, chunk):
        c = torch.randint(0, 256, (chunk, m), device="cuda", dtype=torch.uint8, generator=g)
        pq.append_codes_device(c.data_ptr(), chunk, 0)
        torch.cuda.synchronize()
        host_codes[s:s + chunk] = c.cpu().numpy()
    del c
    assert len(pq) == n
    q = randn(8, d, 17)
    qh = q.cpu().numpy()
    keep_t = (torch.rand(n, device="cuda", generator=g) < 0.01)
    keep = keep_t.cpu().numpy()
    words_t = torch.from_numpy(nat.pack_mask(keep).view(np.int32)).cuda()

    fn = lambda qp, nn, kk, dp, rp: pq.search_device(qp, nn, kk, dp, rp)
    fn_mask = lambda qp, nn, kk, dp, rp: pq.search_device(qp, nn, kk, dp, rp, mask_ptr=words_t.data_ptr())
    dist, rows = search_dev(fn, q, k)
    dmask, rmask = search_dev(fn_mask, q, k)
    check_shape_properties(dist, rows, n)
    check_shape_properties(dmask, rmask, n)
    assert np.all(keep[rmask]), "filtered rows must not be returned"
    # returned distances: bit-exact ADC
    for qi in range(8):
        lut = R.pq_lut(qh[qi], book)
        assert np.array_equal(pq.lut(qh[qi]), lut)
        assert np.array_equal(dist[qi], R.pq_adc(host_codes[rows[qi]], lut))
        assert np.array_equal(dmask[qi], R.pq_adc(host_codes[rmask[qi]], lut))
    # completeness: the oracle over all 100M codes for one query, with and without the filter
    lut = R.pq_lut(qh[2], book)
    acc = lut[0, host_codes[:, 0]].copy()
    for j in range(1, m):
        acc += lut[j, host_codes[:, j]]          # in-place: same fp32 additions, no 400 MB temporaries
    dall = np.sqrt(acc)
    ref_rows = topk_big(dall, k)
    assert np.array_equal(rows[2], ref_rows)
    kept = np.flatnonzero(keep)
    ref_rows_m = kept[topk_big(dall[kept], k)]
    assert np.array_equal(rmask[2], ref_rows_m)
    # single query == batch
    d1, r1 = search_dev(fn, q[5:6].contiguous(), k)
    assert np.array_equal(r1[0], rows[5]) and np.array_equal(d1[0], dist[5])
    pq.close()


# ---------------------------------------------------------------- PQ scan behind a sample bound (5M codes: every query checked in full)
@pytest.mark.parametrize("m", [16, 8])
def test_pq_5m_bound_pass_complete(nat, m):
    """Code sets of 4M rows and more are scanned behind a bound taken from their first rows, and lanes leave a row's
    table sum as soon as it passes the bound.  The oracle scans all rows for every query: equal rows in equal
    order, equal bits, for 1, 2, 3 and 5 queries per call, k = 10 and k = 32, an unselective filter, a filter that
    leaves the sampled prefix without k passing rows, ties (every code row stored twice) and the best rows at the
    very end of the set."""
    n_half, d = 2_600_000, 128
    n = 2 * n_half
    rng = np.random.default_rng(31)
    book = rng.standard_normal((m, 256, d // m), dtype=np.float32)
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    c = torch.randint(0, 256, (n_half, m), device="cuda", dtype=torch.uint8, generator=g)
    q = randn(5, d, 23)
    qh = q.cpu().numpy()
    half = c.cpu().numpy()
    # the rows nearest to query 0: its best codeword in every sub-space, stored last (after both copies)
    lut0 = R.pq_lut(qh[0], book)
    best_code = lut0.argmin(axis=1).astype(np.uint8)
    tail = np.tile(best_code, (7, 1))
    tail[1:, 0] = np.arange(1, 7, dtype=np.uint8) ^ best_code[0]          # six near misses
    host_codes = np.concatenate([half, half, tail])
    n_all = host_codes.shape[0]
    pq = nat.PqCodes(d, m, 256)
    pq.set_codebook(book)
    dev_codes = torch.from_numpy(host_codes).cuda()
    pq.append_codes_device(dev_codes.data_ptr(), n_all, 0)
    torch.cuda.synchronize()
    assert len(pq) == n_all

    keep_half = rng.random(n_all) < 0.5
    keep_late = np.zeros(n_all, bool)
    keep_late[n_all - 400_000:] = rng.random(400_000) < 0.3               # nothing passes in the sampled prefix
    masks = {None: None, "half": keep_half, "late": keep_late}
    words = {name: (None if v is None else torch.from_numpy(nat.pack_mask(v).view(np.int32)).cuda()) for name, v in masks.items()}

    dist_all = []
    for qi in range(5):
        lut = R.pq_lut(qh[qi], book)
        acc = lut[0, host_codes[:, 0]].copy()
        for j in range(1, m):
            acc += lut[j, host_codes[:, j]]          # sequential fp32 sum over the sub-spaces, as pq.py:291-294
        dist_all.append(np.sqrt(acc))

    def oracle(qi, k, keep):
        dall = dist_all[qi]
        if keep is None:
            rows = topk_big(dall, k)
        else:
            kept = np.flatnonzero(keep)
            rows = kept[topk_big(dall[kept], k)]
        return rows, dall[rows]

    ref = {(qi, k, name): oracle(qi, k, masks[name]) for qi in range(5) for k in (10, 32) for name in masks}
    for name in masks:
        mp = 0 if words[name] is None else words[name].data_ptr()
        fn = lambda qp, nn, kk, dp, rp: pq.search_device(qp, nn, kk, dp, rp, mask_ptr=mp)
        for k in (10, 32):
            for lo, hi in ((0, 1), (1, 3), (0, 3), (0, 5)):
                dist, rows = search_dev(fn, q[lo:hi].contiguous(), k)
                for i, qi in enumerate(range(lo, hi)):
                    want_rows, want_d = ref[(qi, k, name)]
                    assert np.array_equal(rows[i], want_rows), (name, k, lo, hi, qi)
                    assert np.array_equal(dist[i], want_d), (name, k, lo, hi, qi)
    # the planted rows are found: the exact match first (lowest of equal rows wins: there is only one)
    assert ref[(0, 10, None)][0][0] == n_all - 7
    pq.close()


# ---------------------------------------------------------------- Flat dot 10M x 768 over 8 shards (configs[2])
def _bf16_round_t(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _dot_atol(rows_h, q_h):
    """Rounding floor of a dot product: a few ulps of sum |c_i q_i| (bounded by max |c|_1 * max |q_i|)."""
    return 1e-5 * float(np.max(np.abs(rows_h).sum(axis=1)) * np.max(np.abs(q_h)))


@pytest.mark.parametrize("storage", ["f32", "f32+filter", "bf16"])
def test_flat_dot_c3_one_shard(nat, storage):
    """One GPU's shard of configs[2]: 1.25M x 768 rows, dot product, batch 1024, k = 10, fp32 and bf16 storage.
    "f32+filter": fp32 storage behind the bf16 filter pass, held to the fp32 bars and to bit equality with the
    search without the filter.
    bf16 storage is checked against the oracle on bf16-rounded rows and queries (products of bf16 values are
    exact in fp32, the sums are fp32): the fp32 bar, 1e-4 relative plus the rounding floor of a 768-term dot
    product.  Recall@10 of the bf16 results against the unrounded fp32 oracle is reported."""
    n, d, nq, k = 1_250_000, 768, 1024, 10
    x = randn(n, d, 4242)
    q = randn(nq, d, 321)
    filtered = storage == "f32+filter"
    if filtered:
        storage = "f32"
    store = nat.RowStore(d, storage=storage)
    store.append_device(x.data_ptr(), n, d, 0)
    fn = lambda qp, m, kk, dp, rp: store.search_device(qp, m, kk, nat.DOT, dp, rp)
    if filtered:
        plain = search_dev(fn, q, k)
        store.set_prefilter("bf16")
    dist, rows = search_dev(fn, q, k)
    assert store.search_stats() == (True, False), "the batch must take the candidate-queue kernel without overflow"
    if filtered:
        assert store.used_prefilter()
        assert np.array_equal(plain[1], rows) and np.array_equal(plain[0].view(np.uint32), dist.view(np.uint32))
    dist2, rows2 = search_dev(fn, q, k)
    assert np.array_equal(rows, rows2) and np.array_equal(dist, dist2), "same call, same bits"
    check_shape_properties(dist, rows, n)

    x_true = x.cpu().numpy()
    q_true = q.cpu().numpy()
    if storage == "bf16":
        xh, qh = _bf16_round_t(x).cpu().numpy(), _bf16_round_t(q).cpu().numpy()
    else:
        xh, qh = x_true, q_true
    del x
    torch.cuda.empty_cache()
    # every returned distance against the oracle's value for that row
    for qi in range(0, nq, 37):
        ref = R.query_distances(qh[qi], xh[rows[qi]], "dot")
        np.testing.assert_allclose(dist[qi], ref, rtol=1e-4, atol=_dot_atol(xh[rows[qi]], qh[qi]))
    # completeness: the oracle over all rows for a few queries
    recall = []
    for qi in (0, 1, 255, 511, 777, 1023):
        ref_all = R.query_distances(qh[qi], xh, "dot")
        ref_rows = R.topk_stable(ref_all, k)
        atol = _dot_atol(xh[ref_rows], qh[qi])
        assert same_ids_or_near_ties(rows[qi], ref_rows, lambda r: ref_all[r], float(ref_all[ref_rows[-1]]),
                                     rtol=1e-4, atol=atol)
        if storage == "bf16":
            true_rows = R.topk_stable(R.query_distances(q_true[qi], x_true, "dot"), k)
            recall.append(len(set(rows[qi].tolist()) & set(true_rows.tolist())) / k)
            # stated tolerance against the unrounded fp32 reference: 2e-2 of the distance scale |q| |c|
            full_true = R.query_distances(q_true[qi], x_true[rows[qi]], "dot")
            scale = np.linalg.norm(q_true[qi]) * np.linalg.norm(x_true[rows[qi]], axis=1)
            assert np.all(np.abs(dist[qi] - full_true) <= 2e-2 * scale)
    if recall:
        print(f"bf16 dot 1.25M x 768: recall@10 vs the fp32 oracle = {np.mean(recall):.3f}")
        assert np.mean(recall) >= 0.8
    # one query at a time agrees with the batch (fp32: the streaming kernel; bf16: the matrix-core kernel with one query)
    for qi in (3, 700):
        d1, r1 = search_dev(fn, q[qi:qi + 1].contiguous(), k)
        ref_all = R.query_distances(qh[qi], xh, "dot")
        kth = float(np.sort(ref_all)[k - 1])
        assert same_ids_or_near_ties(r1[0], rows[qi], lambda r: ref_all[r], kth, rtol=1e-4,
                                     atol=_dot_atol(xh[rows[qi]], qh[qi]))
        np.testing.assert_allclose(d1[0], dist[qi], rtol=1e-4, atol=_dot_atol(xh[rows[qi]], qh[qi]))
    store.close()


@pytest.mark.parametrize("storage", ["f32", "f32+filter", "bf16"])
def test_flat_dot_c3_10m_eight_logical_shards(nat, storage):
    """All of configs[2] on one GPU: 10M x 768 rows held once as a single store and once as 8 row shards.  Every
    shard is searched as rank r of an 8-rank group would search it, its packed block goes where the all-gather
    would put it, and the merge that follows the all-gather must return exactly what the single store returns
    (same rows, same order, same bits).  The returned rows are gathered on the device and checked against the
    oracle; for fp32 the oracle also scans all 10M rows, shard by shard, for three queries."""
    from vectordb_amd.sharded import HipFlatBackend, ShardedFlat, shard_range
    n, d, nq, k, shards = 10_000_000, 768, 1024, 10, 8
    dev = torch.device("cuda", 0)
    q = randn(nq, d, 321)
    # "f32+filter": every store searches behind the bf16 filter pass; held to the fp32 checks below
    prefilter = "bf16" if storage == "f32+filter" else None
    if prefilter:
        storage = "f32"
    qh = (_bf16_round_t(q) if storage == "bf16" else q).cpu().numpy()
    single = HipFlatBackend(d, nat.DOT, storage=storage, prefilter=prefilter)
    parts, backends, blocks = [], [], []
    full_scan_q = (0, 500, 1023)
    ref_scan = {qi: [] for qi in full_scan_q}
    for r in range(shards):
        lo, hi = shard_range(n, shards, r)
        part = randn(hi - lo, d, 9000 + r)
        single.add_device(part)
        be = HipFlatBackend(d, nat.DOT, storage=storage, prefilter=prefilter)
        be.add_device(part)
        torch.cuda.synchronize()
        backends.append(be)
        if storage == "f32":
            parts.append(part)                                   # kept on the device for the row gather below
            part_h = part.cpu().numpy()
            for qi in full_scan_q:
                ref_scan[qi].append(R.query_distances(qh[qi], part_h, "dot"))
            del part_h
        else:
            parts.append(part.to(torch.bfloat16))                 # the rounded values, half the bytes
        del part
    one = ShardedFlat(single, n, 0, 1)
    sd, sr = one.search(q, k)
    torch.cuda.synchronize()
    assert single.store.search_stats() == (True, False)
    assert single.store.used_prefilter() == (prefilter is not None)
    sd_h, sr_h = sd.cpu().numpy(), sr.cpu().numpy()
    check_shape_properties(sd_h, sr_h, n)

    gathered = None
    for r in range(shards):
        as_rank = ShardedFlat(backends[r], n, r, shards)
        local, g, ld, lr, gd, gr, base, od, orow = as_rank._buffers(nq, k, dev)
        backends[r].local_search(q, k, ld, lr)
        blocks.append(local.clone())
        if r == shards - 1:
            gathered, m_gd, m_gr, m_base, m_od, m_orow = g, gd, gr, base, od, orow
    gathered.copy_(torch.cat(blocks))                            # what all_gather_into_tensor leaves on every rank
    backends[-1].merge(m_gd, m_gr, k, m_od, m_orow, m_base)
    torch.cuda.synchronize()
    assert torch.equal(m_orow, sr), "merged shards must return the rows of the single store, in its order"
    assert torch.equal(m_od, sd), "and the same distance bits"

    # returned rows, gathered on the device, against the oracle
    per = n // shards
    for qi in range(0, nq, 61):
        rows_t = torch.stack([parts[int(g) // per][int(g) % per] for g in sr_h[qi]]).to(torch.float32)
        rows_h = rows_t.cpu().numpy()
        ref = R.query_distances(qh[qi], rows_h, "dot")
        np.testing.assert_allclose(sd_h[qi], ref, rtol=1e-4, atol=_dot_atol(rows_h, qh[qi]))
    # completeness against the oracle's scan of all 10M rows (fp32)
    if storage == "f32":
        for qi in full_scan_q:
            ref_all = np.concatenate(ref_scan[qi])
            ref_rows = topk_big(ref_all, k)
            ref_rows_h = torch.stack([parts[int(g) // per][int(g) % per] for g in ref_rows]).cpu().numpy()
            assert same_ids_or_near_ties(sr_h[qi], ref_rows, lambda r: ref_all[r], float(ref_all[ref_rows[-1]]),
                                         rtol=1e-4, atol=_dot_atol(ref_rows_h, qh[qi]))
    single.store.close()
    for be in backends:
        be.store.close()
