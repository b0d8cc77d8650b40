#!/usr/bin/env python3
"""Secondary measurements for the other rows of the hot path (not the driver's
headline bench): bandwidth-bound scans and build steps at BASELINE.json sizes,
each rated against the HBM roofline with its algorithmic bytes.

    python bench_paths.py [--only flat1,l1,c3,bf16,ivf,ivfpq,pq,assign,encode,train,l1batch] [--scale 1.0]

Prints one JSON line per path.  Inputs are synthetic standard-normal (or
uniform random codes) and resident in HBM before timing.
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK = 157.3
MFMA_BF16_PEAK = 2500.0


def timed(fn, steps, warmup=2):
    from vectordb_amd import _native
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    _native.profile_read()
    _native.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    _native.profile_enable(False)
    ms, n = _native.profile_read()
    return wall, (ms / n * 1e-3 if n else float("nan")), n // max(steps, 1)


def emit(name, **kw):
    print(json.dumps({"path": name, **kw}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="flat1,l1,c3,bf16,ivf,ivfpq,pq,assign,encode,train,l1batch")
    ap.add_argument("--scale", type=float, default=1.0, help="scale the row counts (1.0 = BASELINE sizes)")
    ap.add_argument("--cpu", action="store_true",
                    help="also time the NumPy restatement of the reference (oracle/) on a stated down-scaled sample of C3, C4 and C5")
    args = ap.parse_args()
    only = set(args.only.split(","))
    from vectordb_amd import _native as N
    N.init(0)
    dev = torch.device("cuda", 0)
    stream = 0
    gen = torch.Generator(device=dev)

    def randn(n, d, seed):
        gen.manual_seed(seed)
        return torch.randn((n, d), device=dev, dtype=torch.float32, generator=gen)

    def outputs(nq, k):
        return (torch.empty((nq, k), device=dev, dtype=torch.float32), torch.empty((nq, k), device=dev, dtype=torch.int64))

    # ---- Flat, one query at a time (streaming scan), cosine 1M x 384 and Manhattan 1M x 128
    for name, metric, n, d in (("flat1", N.COSINE, int(1_000_000 * args.scale), 384),
                               ("l1", N.MANHATTAN, int(1_000_000 * args.scale), 128)):
        if name not in only:
            continue
        rows = randn(n, d, 42)
        store = N.RowStore(d)
        store.append_device(rows.data_ptr(), n, d, stream)
        torch.cuda.synchronize()
        del rows
        q = randn(1, d, 123)
        od, orow = outputs(1, 10)
        wall, kern, launches = timed(lambda: store.search_device(q.data_ptr(), 1, 10, metric, od.data_ptr(), orow.data_ptr()), 20)
        nbytes = n * d * 4
        emit(name, workload=f"Flat {'cosine' if metric == N.COSINE else 'manhattan'} {n}x{d} fp32, 1 query, k=10",
             qps=round(1 / wall, 1), ms_per_query=round(wall * 1e3, 4), kernel_ms=round(kern * 1e3, 4),
             roofline={"bound": "hbm", "achieved": round(nbytes / kern / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(nbytes / kern / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": nbytes})
        store.close()
        torch.cuda.empty_cache()

    # ---- one GPU's shard of Flat dot-product 10M x 768 over 8 GPUs: 1.25M rows, batch 1024 (MFMA) and batch 1 (HBM)
    if "c3" in only:
        n, d = int(1_250_000 * args.scale), 768
        rows = randn(n, d, 42)
        store = N.RowStore(d)
        store.append_device(rows.data_ptr(), n, d, stream)
        torch.cuda.synchronize()
        del rows
        for nq in (1024, 1):
            q = randn(nq, d, 123)
            od, orow = outputs(nq, 10)
            wall, kern, launches = timed(lambda: store.search_device(q.data_ptr(), nq, 10, N.DOT, od.data_ptr(), orow.data_ptr()), 10)
            if nq == 1:
                nbytes = n * d * 4
                roof = {"bound": "hbm", "achieved": round(nbytes / kern / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(nbytes / kern / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": nbytes}
            else:
                flop = 2.0 * nq * n * d
                roof = {"bound": "mfma", "achieved": round(flop / kern / 1e12, 2), "peak": MFMA_F32_PEAK, "unit": "TFLOP/s",
                        "frac": round(flop / kern / 1e12 / MFMA_F32_PEAK, 4), "flop_per_launch": flop}
            emit("c3", workload=f"Flat dot {n}x{d} fp32 (one of 8 shards of 10M), batch {nq}, k=10",
                 qps=round(nq / wall, 1), ms_per_batch=round(wall * 1e3, 4), kernel_ms=round(kern * 1e3, 4), roofline=roof)
        if args.cpu:
            # the reference's batch path for dot (distance/metrics.py:376-391 then a stable sort per row) on 1/10 of a shard
            from oracle import ref_numpy as R
            ns, nqs = n // 10, 64
            xs = randn(ns, d, 42).cpu().numpy()
            qs = randn(nqs, d, 123).cpu().numpy()
            t0 = time.perf_counter()
            dm = R.pairwise_dot(qs, xs)
            for i in range(nqs):
                R.topk_stable(dm[i], 10)
            dt = time.perf_counter() - t0
            emit("cpu", config="C3", sample=f"Flat dot {ns}x{d} fp32, {nqs} queries as one batch, k=10 (1/10 of one shard)",
                 cores=os.cpu_count(), ms_per_query_at_sample=round(dt / nqs * 1e3, 4),
                 ms_per_query_full_shard_linear=round(dt / nqs * 1e3 * n / ns, 3), kind="port")
        store.close()
        torch.cuda.empty_cache()

    # ---- bf16 storage: configs[1] shape (cosine 1M x 384) and one configs[2] shard (dot 1.25M x 768), batch 1024
    if "bf16" in only:
        for metric, mname, n, d in ((N.COSINE, "cosine", int(1_000_000 * args.scale), 384), (N.DOT, "dot", int(1_250_000 * args.scale), 768)):
            rows = randn(n, d, 42)
            store = N.RowStore(d, storage="bf16")
            store.append_device(rows.data_ptr(), n, d, stream)
            ref = N.RowStore(d)
            ref.append_device(rows.data_ptr(), n, d, stream)
            torch.cuda.synchronize()
            del rows
            nq = 1024
            q = randn(nq, d, 123)
            od, orow = outputs(nq, 10)
            rd, rrow = outputs(nq, 10)
            ref.search_device(q.data_ptr(), nq, 10, metric, rd.data_ptr(), rrow.data_ptr())
            wall, kern, launches = timed(lambda: store.search_device(q.data_ptr(), nq, 10, metric, od.data_ptr(), orow.data_ptr()), 10)
            a, b = orow.cpu().numpy(), rrow.cpu().numpy()
            recall = float(np.mean([len(set(a[i]) & set(b[i])) / 10.0 for i in range(nq)]))
            flop = 2.0 * nq * n * d
            emit("bf16", workload=f"Flat {mname} {n}x{d} bf16 storage, batch {nq}, k=10", qps=round(nq / wall, 1),
                 ms_per_batch=round(wall * 1e3, 4), kernel_ms=round(kern * 1e3, 4), recall_at_10_vs_fp32=round(recall, 4),
                 roofline={"bound": "mfma", "achieved": round(flop / kern / 1e12, 2), "peak": MFMA_BF16_PEAK, "unit": "TFLOP/s",
                           "frac": round(flop / kern / 1e12 / MFMA_BF16_PEAK, 4), "flop_per_launch": flop})
            # the fp32 store behind the bf16 filter pass: the same 256 x 256 kernel, fp32 results bit for bit
            ref.set_prefilter("bf16")
            fd, frow = outputs(nq, 10)
            wall, kern, launches = timed(lambda: ref.search_device(q.data_ptr(), nq, 10, metric, fd.data_ptr(), frow.data_ptr()), 10)
            same = bool(torch.equal(fd.view(torch.int32), rd.view(torch.int32)) and torch.equal(frow, rrow))
            emit("prefilter", workload=f"Flat {mname} {n}x{d} fp32 storage behind the bf16 filter pass, batch {nq}, k=10",
                 qps=round(nq / wall, 1), ms_per_batch=round(wall * 1e3, 4), kernel_ms=round(kern * 1e3, 4),
                 identical_to_fp32_search=same, used_filter=ref.used_prefilter(),
                 roofline={"bound": "mfma", "achieved": round(flop / kern / 1e12, 2), "peak": MFMA_BF16_PEAK, "unit": "TFLOP/s",
                           "frac": round(flop / kern / 1e12 / MFMA_BF16_PEAK, 4), "flop_per_launch": flop})
            store.close(); ref.close()
            torch.cuda.empty_cache()

    # ---- nearest-centroid assignment: n x 4096 x 128 (MFMA)
    if "assign" in only:
        n, k, d = int(1_048_576 * args.scale), 4096, 128
        x = randn(n, d, 42)
        cent = N.RowStore(d)
        cent.append_device(x.data_ptr(), k, d, stream)
        od, orow = outputs(n, 1)
        wall, kern, _ = timed(lambda: cent.search_device(x.data_ptr(), n, 1, N.EUCLIDEAN, od.data_ptr(), orow.data_ptr()), 5)
        flop = 2.0 * n * k * d
        emit("assign", workload=f"k-means assignment {n} rows x {k} centroids x {d}", rows_per_s=round(n / wall, 1),
             ms=round(wall * 1e3, 3), kernel_ms=round(kern * 1e3, 3),
             roofline={"bound": "mfma", "achieved": round(flop / kern / 1e12, 2), "peak": MFMA_F32_PEAK, "unit": "TFLOP/s",
                       "frac": round(flop / kern / 1e12 / MFMA_F32_PEAK, 4), "flop_per_launch": flop})
        cent.close()
        del x, od, orow
        torch.cuda.empty_cache()

    # ---- IVF training as BASELINE configs[3] states it: k-means++ and 20 Lloyd steps over 256 x 4096 = 1 048 576
    #      sampled rows, 4096 clusters, seed 42, through IVFIndex.train (rows uploaded once, everything else on the device)
    if "train" in only:
        from vectordb_amd.index import IVFIndex
        n, d, nlist = int(1_048_576 * args.scale), 128, max(8, int(4096 * args.scale))
        rng = np.random.default_rng(42)
        x = rng.standard_normal((n, d), dtype=np.float32)
        index = IVFIndex(dimension=d, n_clusters=nlist, n_probe=32, n_iter=20, seed=42)
        t0 = time.perf_counter()
        index.train(x)
        wall = time.perf_counter() - t0
        ph = index._last_kmeans.phase_seconds
        steps = max(1, ph["lloyd_steps"])
        emit("train", workload=f"IVFIndex.train: k-means++ + Lloyd, {n} x {d} rows, {nlist} clusters, n_iter=20, seed=42",
             seconds=round(wall, 3), upload_s=round(ph["upload"], 3), init_pp_s=round(ph["init_pp"], 3),
             lloyd_s=round(ph["lloyd"], 3), lloyd_steps=ph["lloyd_steps"], ms_per_lloyd_step=round(ph["lloyd"] / steps * 1e3, 2),
             assign_tflops_if_all_of_a_step=round(2.0 * n * nlist * d / (ph["lloyd"] / steps) / 1e12, 1))
        del x, index
        torch.cuda.empty_cache()

    # ---- Manhattan with a real batch: 1M x 128, 1024 queries (no matrix form; the streaming kernel, 8 queries per row chunk)
    if "l1batch" in only:
        n, d, nq = int(1_000_000 * args.scale), 128, 1024
        rows = randn(n, d, 42)
        store = N.RowStore(d)
        store.append_device(rows.data_ptr(), n, d, stream)
        torch.cuda.synchronize()
        del rows
        q = randn(nq, d, 123)
        od, orow = outputs(nq, 10)
        wall, kern, launches = timed(lambda: store.search_device(q.data_ptr(), nq, 10, N.MANHATTAN, od.data_ptr(), orow.data_ptr()), 5)
        pair_ops = 3.0 * nq * n * d                  # subtract, abs, add per (row, query, dimension)
        emit("l1batch", workload=f"Flat manhattan {n}x{d} fp32, batch {nq}, k=10", qps=round(nq / wall, 1),
             ms_per_batch=round(wall * 1e3, 3), kernel_ms_avg=round(kern * 1e3, 3), kernels_per_batch=launches,
             valu_tops=round(pair_ops / wall / 1e12, 2))
        store.close()
        torch.cuda.empty_cache()

    # ---- IVF nlist=4096 nprobe=32 over 10M x 128
    if "ivf" in only:
        n, d, nlist, nprobe = int(10_000_000 * args.scale), 128, 4096, 32
        rng = np.random.default_rng(42)
        x = rng.standard_normal((n, d), dtype=np.float32)
        # a short k-means on a sample (device assignment + accumulation) so that the lists are balanced like trained ones
        sample = x[rng.choice(n, min(n, 64 * nlist), replace=False)]
        cent = sample[rng.choice(len(sample), nlist, replace=False)].copy()
        t0 = time.perf_counter()
        for _ in range(6):
            a, _d = N.assign(sample, cent, N.EUCLIDEAN)
            sums, counts = N.kmeans_accumulate(sample, a, nlist)
            nz = counts > 0
            cent[nz] = (sums[nz] / counts[nz, None]).astype(np.float32)
        t_train = time.perf_counter() - t0
        t0 = time.perf_counter()
        assign, _ = N.assign(x, cent, N.EUCLIDEAN)
        t_assign = time.perf_counter() - t0
        lists = N.IvfLists(d, N.EUCLIDEAN)
        lists.set_centroids(cent)
        t0 = time.perf_counter()
        lists.build(x, assign)
        t_build = time.perf_counter() - t0
        for nq in (1, 1024):
            q = randn(nq, d, 123)
            od, orow = outputs(nq, 10)
            wall, kern, launches = timed(lambda: lists.search_device(q.data_ptr(), nq, 10, nprobe, od.data_ptr(), orow.data_ptr()), 10)
            scanned = lists.last_scan_rows()
            nbytes = (nq * nlist + scanned) * d * 4
            # a batch shares lists between its queries: HBM hands over at most every list and centroid once
            unique = min(nbytes, (n + nlist) * d * 4)
            emit("ivf", workload=f"IVF nlist={nlist} nprobe={nprobe}, {n}x{d}, batch {nq}, k=10",
                 qps=round(nq / wall, 1), ms_per_batch=round(wall * 1e3, 4), kernel_ms_avg=round(kern * 1e3, 4),
                 kernels_per_batch=launches, rows_scanned_per_query=scanned // nq,
                 train_s=round(t_train, 2), assign_s=round(t_assign, 2), build_s=round(t_build, 2),
                 longest_list=int(np.bincount(assign, minlength=nlist).max()),
                 roofline={"bound": "hbm", "achieved": round(unique / wall / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(unique / wall / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_batch": unique,
                           "per_query_bytes_times_queries": nbytes,
                           "note": "bytes HBM must deliver once (every probed list and the centroids, shared by the batch) "
                                   "/ whole-batch wall time (probe + scan + select); per-query bytes x queries is listed beside it"})
        if args.cpu:
            # IVFIndex.search as the reference runs it (ivf.py:607-632), on the lists the first million rows fall into
            from oracle import ref_numpy as R
            ns, nqs = min(n, 1_000_000), 8
            sub_lists = R.ivf_build_lists(assign[:ns].tolist(), nlist)
            qs = randn(nqs, d, 123).cpu().numpy()
            t0 = time.perf_counter()
            scanned_cpu = 0
            for i in range(nqs):
                R.ivf_search(qs[i], x[:ns], cent, sub_lists, 10, nprobe, "euclidean")
                scanned_cpu += sum(len(sub_lists[c]) for c in R.ivf_probe_order(qs[i], cent, "euclidean", nprobe))
            dt = time.perf_counter() - t0
            emit("cpu", config="C4", sample=f"IVF nlist={nlist} nprobe={nprobe} over the first {ns} of {n} rows, {nqs} single queries, k=10",
                 cores=os.cpu_count(), ms_per_query_at_sample=round(dt / nqs * 1e3, 3), rows_scanned_per_query=scanned_cpu // nqs,
                 ms_per_query_full_linear_in_rows_scanned=round(dt / nqs * 1e3 * n / ns, 2), kind="port",
                 note="vectorised NumPy restatement; the reference itself loops in Python over every member (about 7 us per row)")
        lists.close()
        del x
        torch.cuda.empty_cache()

    # ---- IVF-PQ (HybridIndex): nlist=4096 nprobe=32 over 10M x 128 rows stored as 16-byte codes
    if "ivfpq" in only:
        n, d, m, nlist, nprobe = int(10_000_000 * args.scale), 128, 16, 4096, 32
        rng = np.random.default_rng(42)
        x = rng.standard_normal((n, d), dtype=np.float32)
        sample = x[rng.choice(n, min(n, 64 * nlist), replace=False)]
        cent = sample[rng.choice(len(sample), nlist, replace=False)].copy()
        for _ in range(4):
            a, _d = N.assign(sample, cent, N.EUCLIDEAN_SQ)
            sums, counts = N.kmeans_accumulate(sample, a, nlist)
            nz = counts > 0
            cent[nz] = (sums[nz] / counts[nz, None]).astype(np.float32)
        book = (0.3 * rng.standard_normal((m, 256, d // m))).astype(np.float32)      # untrained codebook: the scan cost is the same
        h = N.IvfPqLists(d, m, 256)
        h.set_quantizers(cent, book)
        t0 = time.perf_counter()
        lists, codes = h.encode(x)
        t_encode = time.perf_counter() - t0
        del x
        order = np.argsort(lists, kind="stable")
        offs = np.searchsorted(lists[order], np.arange(nlist + 1))
        t0 = time.perf_counter()
        h.build(codes, order, offs)
        t_build = time.perf_counter() - t0
        for nq in (1, 1024):
            q = randn(nq, d, 123)
            od, orow = outputs(nq, 10)
            wall, kern, launches = timed(lambda: h.search_device(q.data_ptr(), nq, 10, nprobe, od.data_ptr(), orow.data_ptr()), 10)
            emit("ivfpq", workload=f"IVF-PQ nlist={nlist} nprobe={nprobe} m={m}, {n}x{d}, batch {nq}, k=10",
                 qps=round(nq / wall, 1), ms_per_batch=round(wall * 1e3, 4), kernel_ms_avg=round(kern * 1e3, 4),
                 kernels_per_batch=launches, encode_rows_per_s=round(n / t_encode, 1), build_s=round(t_build, 2))
        h.close()
        torch.cuda.empty_cache()

    # ---- PQ m=16 nbits=8 ADC scan over 100M codes with a filter bitmask
    if "pq" in only:
        n, m, d = int(100_000_000 * args.scale), 16, 128
        pq = N.PqCodes(d, m, 256)
        rng = np.random.default_rng(7)
        pq.set_codebook(rng.standard_normal((m, 256, d // m), dtype=np.float32))
        chunk = 10_000_000
        for s in range(0, n, chunk):
            c = min(chunk, n - s)
            codes = torch.randint(0, 256, (c, m), device=dev, dtype=torch.uint8)
            pq.append_codes_device(codes.data_ptr(), c, stream)
            torch.cuda.synchronize()
        del codes
        for sel in (None, 0.5, 0.01):
            mask = None
            if sel is not None:
                bits = (torch.rand(((n + 31) // 32) * 32, device=dev) < sel).view(-1, 32)
                weights = (2 ** torch.arange(32, device=dev, dtype=torch.int64))
                mask = (bits.to(torch.int64) * weights).sum(1).to(torch.int32).contiguous()   # wraps: same 32 bits
                del bits
            for nq in (1, 8):
                q = randn(nq, d, 123)
                od, orow = outputs(nq, 10)
                mp = mask.data_ptr() if mask is not None else 0
                wall, kern, launches = timed(lambda: pq.search_device(q.data_ptr(), nq, 10, od.data_ptr(), orow.data_ptr(), mask_ptr=mp), 10)
                # queries that share one pass over the codes: eight on the one-byte tables, four on the fp32 tables
                passes = (nq + 7) // 8 if pq.search_stats()[0] else (nq + 3) // 4
                nbytes = passes * (n * m + (n // 8 if mask is not None else 0))
                emit("pq", workload=f"PQ m={m} nbits=8 ADC scan, {n} codes, batch {nq}, filter={sel}", byte_tables=pq.search_stats()[0],
                     qps=round(nq / wall, 2), ms_per_batch=round(wall * 1e3, 4), kernel_ms=round(kern * 1e3, 4),
                     roofline={"bound": "hbm", "achieved": round(nbytes / kern / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(nbytes / kern / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": nbytes})
        if args.cpu:
            # PQIndex.search (pq.py:268-301): table, sequential fp32 sum over the sub-spaces, sqrt, stable top-k; 2M codes
            from oracle import ref_numpy as R
            ns, nqs = min(n, 2_000_000), 4
            rng_c = np.random.default_rng(11)
            codes_s = rng_c.integers(0, 256, size=(ns, m), dtype=np.uint8)
            book = np.random.default_rng(7).standard_normal((m, 256, d // m), dtype=np.float32)
            qs = randn(nqs, d, 123).cpu().numpy()
            t0 = time.perf_counter()
            for i in range(nqs):
                R.pq_search(qs[i], book, codes_s, 10)
            dt = time.perf_counter() - t0
            emit("cpu", config="C5", sample=f"PQ m={m} nbits=8 ADC scan over {ns} of {n} codes, {nqs} single queries, k=10",
                 cores=os.cpu_count(), ms_per_query_at_sample=round(dt / nqs * 1e3, 3),
                 ms_per_query_full_linear=round(dt / nqs * 1e3 * n / ns, 1), kind="port",
                 note="vectorised NumPy restatement; the reference's own scan is a Python loop (about 7 us per row)")
        pq.close()
        torch.cuda.empty_cache()

    # ---- PQ encode throughput
    if "encode" in only:
        n, m, d = int(1_000_000 * args.scale), 16, 128
        rng = np.random.default_rng(3)
        x = rng.standard_normal((n, d), dtype=np.float32)
        book = rng.standard_normal((m, 256, d // m), dtype=np.float32)
        N.pq_encode(x[:1000], book)
        t0 = time.perf_counter()
        N.pq_encode(x, book)
        dt = time.perf_counter() - t0
        emit("encode", workload=f"PQ encode {n}x{d}, m={m}, host buffers in and out", rows_per_s=round(n / dt, 1), s=round(dt, 3))


if __name__ == "__main__":
    main()
