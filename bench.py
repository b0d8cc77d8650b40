#!/usr/bin/env python3
"""Headline benchmark: Flat index, cosine, 1M x 384 fp32 rows, one batch of
1024 queries per step, k = 10 (BASELINE.json configs[1]), through the C ABI.

    python bench.py --gpus N --steps K --warmup W

One process per GPU.  With N > 1 and no WORLD_SIZE in the environment the
script starts `python -m torch.distributed.run --nproc-per-node N` on itself
as a child process (before anything touches a GPU), relays its output and
exits with its code; under torch.distributed.run it reads RANK / LOCAL_RANK /
WORLD_SIZE as usual.

The rows are generated in fixed blocks of global rows, each block from its own
seed, so every world size searches the SAME 1M rows.  With N > 1 they are
sharded by contiguous row blocks (strong scaling: total work per step is
fixed), every rank searches all 1024 queries against its shard, and the
per-rank top-10 lists are merged after one all-gather (RCCL).  Rows and queries
are resident in HBM before the timed region starts; a step is: query norms ->
a strided 1/64 sample pass that bounds each query's 10th best distance -> the
MFMA distance kernel over all rows, which queues the few survivors of that
bound -> per-query selection of the queue [-> all-gather -> merge].

Rank 0 prints one JSON line.  `roofline` rates the dominant kernel
(flat_queue_fast_kernel, fp32 MFMA) by 2*Q*N_local*D flop per launch over its
HIP-event time; `cpu_baseline` (N = 1) times the NumPy oracle (the reference's
arithmetic) on a sample of the same queries and rows on the host cores; a
sample of the queries is checked against that oracle at every N.

`extra` holds side measurements that do not enter `value`:
  prefilter the headline's steps on the same store behind the bf16 filter pass
            (vdb_rows_set_prefilter): queries/s, the bf16 kernel's time and
            fraction of the bf16 MFMA peak, and whether distances and rows are
            bit for bit the headline's;
  c3        BASELINE.json configs[2]: Flat dot 10M x 768, rows / N per GPU,
            fp32 storage, fp32 storage behind the bf16 filter pass (compared
            bit for bit with the former) and bf16 storage, batch 1024
            (queries/s, kernel ms, fraction of the MFMA peak of the kernel's
            type), with a spot check of the returned distances against the oracle;
  c1        configs[0]: 10k x 128 euclidean, 100 single queries, through the
            Collection.search-shaped path and as bare distances + top-k, on the
            GPU and on the host oracle (N = 1 only);
  ceilings  a device-to-device copy and a register-only MFMA loop measured on
            this GPU, beside the spec peaks the fractions are quoted against.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_ROWS, DIM, N_QUERIES, TOP_K = 1_000_000, 384, 1024, 10
C3_ROWS, C3_DIM = 10_000_000, 768
MFMA_F32_PEAK_TFLOPS = 157.3          # MI355X dense fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0        # MI355X dense bf16 matrix peak
HBM_PEAK_GBS = 8000.0
GEN_BLOCK = 15625                      # rows per generation block: 1M = 64 blocks, 10M = 640 blocks


def host_threads() -> int:
    try:
        from threadpoolctl import threadpool_info
        n = [p.get("num_threads", 0) for p in threadpool_info() if p.get("user_api") == "blas"]
        if n:
            return int(max(n))
    except Exception:
        pass
    return int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))


def launch_ranks(n: int) -> int:
    """Start n ranks of this script under torch.distributed.run as a child process.
    Called before any GPU call of this process (device_count does not initialise the GPU)."""
    visible = torch.cuda.device_count()
    if visible < n:
        print(f"bench.py: --gpus {n} asked, {visible} GPU(s) visible on this node", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def block_rows(lo: int, hi: int, d: int, seed: int, dev, gen) -> "torch.Tensor":
    """Global rows [lo, hi) of the synthetic set: standard normal, block b of GEN_BLOCK rows from seed + b."""
    parts = []
    for b in range(lo // GEN_BLOCK, (hi + GEN_BLOCK - 1) // GEN_BLOCK):
        gen.manual_seed(seed + b)
        blk = torch.randn((GEN_BLOCK, d), device=dev, dtype=torch.float32, generator=gen)
        b0 = b * GEN_BLOCK
        parts.append(blk[max(lo - b0, 0):min(hi - b0, GEN_BLOCK)])
    return parts[0] if len(parts) == 1 else torch.cat(parts)


def fill_store(backend, lo: int, hi: int, d: int, seed: int, dev, gen, chunk_rows: int = 8 * GEN_BLOCK) -> None:
    """Append global rows [lo, hi) to the backend's row store, a few blocks at a time."""
    for c0 in range(lo, hi, chunk_rows):
        part = block_rows(c0, min(hi, c0 + chunk_rows), d, seed, dev, gen).contiguous()
        backend.add_device(part)
        torch.cuda.synchronize()
        del part


def same_ids_or_near_ties(got, want, ref_dist_of_row, kth, rtol, atol) -> bool:
    # ids may differ only where the oracle's own distance ties its k-th distance to rounding
    return all(abs(float(ref_dist_of_row(r)) - kth) <= rtol * abs(kth) + atol
               for r in set(map(int, got)) ^ set(map(int, want)))


def timed_steps(index, queries, k, steps, warmup, barrier, native):
    """warmup untimed steps, then `steps` timed ones between barriers; returns (seconds, kernel ms, launches, out)."""
    out = None
    for _ in range(warmup):
        out = index.search(queries, k)
    torch.cuda.synchronize()
    barrier()
    native.profile_read()
    native.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = index.search(queries, k)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    native.profile_enable(False)
    kernel_ms, launches = native.profile_read()
    return elapsed, kernel_ms, launches, out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=N_ROWS)
    ap.add_argument("--dim", type=int, default=DIM)
    ap.add_argument("--queries", type=int, default=N_QUERIES)
    ap.add_argument("--cpu-sample", type=int, default=48, help="queries timed on the host oracle at N = 1 (0 = skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements (c3, c1, ceilings)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=dev)

    from vectordb_amd import _native
    from vectordb_amd.sharded import HipFlatBackend, ShardedFlat, shard_range
    _native.init(local_rank)

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    n_total, d, nq, k = args.rows, args.dim, args.queries, TOP_K
    lo, hi = shard_range(n_total, world, rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(123)
    queries = torch.randn((nq, d), device=dev, dtype=torch.float32, generator=gen)

    backend = HipFlatBackend(d, _native.COSINE, prefilter=None)      # the fp32 kernels alone: the headline
    fill_store(backend, lo, hi, d, 42, dev, gen)
    index = ShardedFlat(backend, n_total, rank, world)

    elapsed, kernel_ms, launches, (out_d, out_r) = timed_steps(index, queries, k, args.steps, args.warmup, barrier, _native)
    used_queue, _ = backend.store.search_stats()          # which kernel the library took for this shard size
    elapsed = max_over_ranks(elapsed)
    ms_per_step = elapsed / args.steps * 1e3
    qps = nq * args.steps / elapsed

    # ---- roofline of the dominant kernel (this rank's launches)
    flop_per_launch = 2.0 * nq * (hi - lo) * d
    kern_ms = kernel_ms / max(launches, 1)
    achieved = flop_per_launch / (kern_ms * 1e-3) / 1e12 if launches else 0.0
    traffic = None          # HBM bytes per launch from the committed rocprofv3 PMC passes (same workload), if present
    try:
        if (n_total, d, nq, world) == (N_ROWS, DIM, N_QUERIES, 1):
            traffic = json.load(open(os.path.join(ROOT, "profiles", "bench_traffic.json")))["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic,
                "kernel": "flat_queue_fast_kernel<cosine>" if used_queue else "flat_gemm_kernel<cosine>", "kernel_ms": round(kern_ms, 4), "launches": launches,
                "flop_per_launch": flop_per_launch}

    # ---- parity sample (rank 0, every N) and CPU baseline (N = 1): the oracle over ALL rows for a few queries
    cpu_baseline = None
    parity = None
    out_d_h, out_r_h = out_d.cpu().numpy(), out_r.cpu().numpy()
    s = min(args.cpu_sample if world == 1 else min(args.cpu_sample, 8), nq)
    if rank == 0 and s > 0:
        from oracle import ref_numpy as R
        rows_host = np.empty((n_total, d), np.float32)
        step = 8 * GEN_BLOCK
        for c0 in range(0, n_total, step):
            c1 = min(n_total, c0 + step)
            rows_host[c0:c1] = block_rows(c0, c1, d, 42, dev, gen).cpu().numpy()
        q_host = queries[:s].cpu().numpy()
        R.flat_search(q_host[0], rows_host[:1000], k, "cosine")           # warm the BLAS threads
        t0 = time.perf_counter()
        ref = [R.flat_search(q_host[i], rows_host, k, "cosine") for i in range(s)]
        cpu_s = time.perf_counter() - t0
        ok = True
        for i in range(min(s, 16)):          # the near-tie check recomputes full distance rows: 16 queries are enough
            full = R.query_distances(q_host[i], rows_host, "cosine")
            ok &= same_ids_or_near_ties(out_r_h[i], ref[i][0], lambda r: full[r], float(ref[i][1][-1]), rtol=1e-4, atol=2e-6)
            ok &= bool(np.allclose(out_d_h[i], ref[i][1], rtol=1e-4, atol=2e-6))
        parity = bool(ok)
        if world == 1:
            cpu_baseline = {"value": round(s / cpu_s, 3), "unit": "queries/s", "cores": host_threads(), "kind": "port",
                            "sample": f"{s} of the {nq} queries, one at a time, against all {n_total} rows "
                                      f"(NumPy restatement of query_distances + stable top-{k})"}
        del rows_host

    # ---- the same steps behind the bf16 filter pass (vdb_rows_set_prefilter): bit-equal results, not part of `value`
    prefilter = None
    if not args.no_extra:
        try:
            backend.store.set_prefilter("bf16")
            p_elapsed, p_kernel_ms, p_launches, (p_d, p_r) = timed_steps(index, queries, k, args.steps, args.warmup, barrier, _native)
            p_elapsed = max_over_ranks(p_elapsed)
            p_kern = p_kernel_ms / max(p_launches, 1)
            p_tflops = flop_per_launch / (p_kern * 1e-3) / 1e12 if p_launches else 0.0
            prefilter = {"what": "same workload and steps; the pass over all rows on bf16 copies with a loosened bound, the fp32 "
                                 "dot products of the remaining rows recomputed (results bit-equal to the headline's)",
                         "queries_per_s": round(nq * args.steps / p_elapsed, 1), "ms_per_step": round(p_elapsed / args.steps * 1e3, 4),
                         "kernel": "flat_queue_wide_kernel<cosine> (bf16 MFMA)", "kernel_ms": round(p_kern, 4),
                         "tflops": round(p_tflops, 1), "peak_tflops": MFMA_BF16_PEAK_TFLOPS,
                         "frac": round(p_tflops / MFMA_BF16_PEAK_TFLOPS, 4), "used_filter": backend.store.used_prefilter(),
                         "identical_to_fp32_search": bool(torch.equal(p_d.view(torch.int32), out_d.view(torch.int32))
                                                          and torch.equal(p_r, out_r))}
        except Exception as e:
            prefilter = {"error": f"{type(e).__name__}: {e}"}

    backend.store.close()
    del index, backend
    torch.cuda.empty_cache()

    extra = None
    if not args.no_extra:
        extra = {"prefilter": prefilter}
        try:
            extra["c3"] = bench_c3(world, rank, dev, gen, barrier, max_over_ranks, _native)
        except Exception as e:          # a side measurement must not take the headline line with it
            extra["c3"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1:
            for name, fn in (("c1", bench_c1), ("ceilings", bench_ceilings)):
                try:
                    extra[name] = fn(dev, _native)
                except Exception as e:
                    extra[name] = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        print(json.dumps({
            "metric": "queries/sec at k=10 (exact), Flat cosine",
            "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Flat cosine 1M x 384 fp32, batch 1024 queries, k=10 (BASELINE configs[1])",
                       "rows": n_total, "dim": d, "queries_per_step": nq, "k": k,
                       "sharding": f"rows/{world}" + (" + all-gather top-k merge" if world > 1 else "")},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity_sample_ok": parity,
            # the same steps with the pass over all rows on bf16 copies and the candidates recomputed in fp32
            # (bit-equal results; details in extra.prefilter); not the headline, whose kernel is the fp32 one
            "value_behind_bf16_filter": (prefilter or {}).get("queries_per_s"),
            "identical_behind_bf16_filter": (prefilter or {}).get("identical_to_fp32_search"),
            "extra": extra,
        }))
    if world > 1:
        dist.destroy_process_group()


def bench_c3(world, rank, dev, gen, barrier, max_over_ranks, native, steps=5, warmup=2):
    """BASELINE.json configs[2]: Flat dot-product over 10M x 768 rows, rows / world per GPU, batch 1024, k = 10,
    fp32 and bf16 storage.  Same step as the headline: local search [-> all-gather -> merge]."""
    from oracle import ref_numpy as R
    from vectordb_amd.sharded import HipFlatBackend, ShardedFlat, shard_range
    n_total, d, nq, k = C3_ROWS, C3_DIM, N_QUERIES, TOP_K
    lo, hi = shard_range(n_total, world, rank)
    gen.manual_seed(321)
    queries = torch.randn((nq, d), device=dev, dtype=torch.float32, generator=gen)
    out = {"workload": f"Flat dot 10M x 768, batch {nq}, k={k}, rows/{world} per GPU (BASELINE configs[2])",
           "rows_per_gpu": hi - lo, "steps": steps, "warmup": warmup}
    f32_result = None
    for name, storage, peak in (("f32", "f32", MFMA_F32_PEAK_TFLOPS), ("f32_bf16_filter", "f32", MFMA_BF16_PEAK_TFLOPS),
                                ("bf16", "bf16", MFMA_BF16_PEAK_TFLOPS)):
        backend = HipFlatBackend(d, native.DOT, storage=storage, prefilter="bf16" if name == "f32_bf16_filter" else None)
        fill_store(backend, lo, hi, d, 4242, dev, gen)
        index = ShardedFlat(backend, n_total, rank, world)
        elapsed, kernel_ms, launches, (od, orow) = timed_steps(index, queries, k, steps, warmup, barrier, native)
        elapsed = max_over_ranks(elapsed)
        kern = kernel_ms / max(launches, 1)
        tflops = 2.0 * nq * (hi - lo) * d / (kern * 1e-3) / 1e12 if launches else 0.0
        # spot check: the returned rows of a few queries, regenerated from their blocks, against the oracle's
        # distances (on bf16-rounded values for bf16 storage); 1e-4 relative + the dot form's rounding floor
        ok = None
        if rank == 0:
            ok = True
            od_h, or_h = od.cpu().numpy(), orow.cpu().numpy()
            qh = queries.cpu().numpy()
            for qi in (0, nq // 2, nq - 1):
                rows = np.stack([block_rows(int(r), int(r) + 1, d, 4242, dev, gen)[0].cpu().numpy() for r in or_h[qi]])
                qv = qh[qi]
                if storage == "bf16":
                    rows = torch.from_numpy(rows).to(torch.bfloat16).to(torch.float32).numpy()
                    qv = torch.from_numpy(qv).to(torch.bfloat16).to(torch.float32).numpy()
                ref = R.query_distances(qv, rows, "dot")
                atol = 1e-5 * float(np.max(np.abs(rows).sum(axis=1)) * np.max(np.abs(qv)))
                ok &= bool(np.allclose(od_h[qi], ref, rtol=1e-4, atol=atol)) and bool(np.all(np.diff(od_h[qi]) >= 0))
        out[name] = {"queries_per_s": round(nq * steps / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 4),
                     "kernel_ms": round(kern, 4), "tflops": round(tflops, 1), "peak_tflops": peak,
                     "frac": round(tflops / peak, 4), "returned_distances_ok": ok}
        if name == "f32":
            f32_result = (od.clone(), orow.clone())
        elif name == "f32_bf16_filter":
            out[name]["identical_to_f32"] = bool(torch.equal(od.view(torch.int32), f32_result[0].view(torch.int32))
                                                 and torch.equal(orow, f32_result[1]))
            f32_result = None
        backend.store.close()
        del index, backend
        torch.cuda.empty_cache()
    return out


def bench_c1(dev, native, n=10_000, d=128, nq=100, k=10):
    """BASELINE.json configs[0]: 10k x 128 euclidean, 100 queries one at a time, k = 10.  Two shapes on each side:
    the whole Collection.search call (validation, search, ids, scores, result objects) and the bare kernel
    (distances + top-k on raw arrays).  CPU = the NumPy oracle of the same two shapes on the host cores."""
    from oracle import ref_numpy as R
    from vectordb_amd.core.collection import Collection
    rng = np.random.RandomState(42)
    x = rng.randn(n, d).astype(np.float32)
    rng = np.random.RandomState(123)
    q = rng.randn(nq, d).astype(np.float32)
    ids = [f"vec_{i}" for i in range(n)]

    col = Collection("bench", dimension=d, metric="euclidean")
    col.add_batch([{"id": ids[i], "vector": x[i]} for i in range(n)])
    col.search(q[0], k)
    t0 = time.perf_counter()
    got = [col.search(q[i], k) for i in range(nq)]
    gpu_collection = nq / (time.perf_counter() - t0)

    store = native.RowStore(d)
    store.append(x)
    store.search(q[:1], k, native.EUCLIDEAN)
    t0 = time.perf_counter()
    bare = [store.search(q[i:i + 1], k, native.EUCLIDEAN) for i in range(nq)]
    gpu_bare = nq / (time.perf_counter() - t0)
    store.close()

    t0 = time.perf_counter()
    ref = [R.collection_search(q[i], x, ids, k, "euclidean") for i in range(nq)]
    cpu_collection = nq / (time.perf_counter() - t0)
    t0 = time.perf_counter()
    ref_bare = [R.flat_search(q[i], x, k, "euclidean") for i in range(nq)]
    cpu_bare = nq / (time.perf_counter() - t0)

    ok = True
    for i in range(nq):
        want_rows, want_d = ref_bare[i]
        full = None
        got_rows = bare[i][1][0]
        if set(map(int, got_rows)) != set(map(int, want_rows)):
            full = R.query_distances(q[i], x, "euclidean")
            ok &= same_ids_or_near_ties(got_rows, want_rows, lambda r: full[r], float(want_d[-1]), 1e-4, 1e-5)
        ok &= bool(np.allclose(bare[i][0][0], want_d, rtol=1e-4, atol=1e-4))
        ok &= [r.id for r in got[i]] == [ids[int(r)] for r in got_rows] or full is not None
    return {"workload": f"Flat euclidean {n} x {d}, {nq} single queries, k={k} (BASELINE configs[0])",
            "gpu_collection_search_qps": round(gpu_collection, 1), "gpu_bare_search_qps": round(gpu_bare, 1),
            "cpu_collection_search_qps": round(cpu_collection, 1), "cpu_bare_search_qps": round(cpu_bare, 1),
            "cpu_cores": host_threads(), "cpu_kind": "port", "results_match_oracle": bool(ok)}


def bench_ceilings(dev, native):
    """What this GPU reaches on the two limits the fractions are quoted against: a large device-to-device copy
    (read + write bytes per second) and register-only MFMA loops (no memory traffic at all)."""
    copy_gbs, f32_tflops, bf16_tflops, valu_tops = native.measure_ceilings()
    return {"copy_gbs_read_plus_write": round(copy_gbs, 1), "hbm_spec_gbs": HBM_PEAK_GBS,
            "mfma_f32_loop_tflops": round(f32_tflops, 1), "mfma_f32_spec_tflops": MFMA_F32_PEAK_TFLOPS,
            "mfma_bf16_loop_tflops": round(bf16_tflops, 1), "mfma_bf16_spec_tflops": MFMA_BF16_PEAK_TFLOPS,
            "valu_direct_form_loop_tops": round(valu_tops, 1)}


if __name__ == "__main__":
    main()
