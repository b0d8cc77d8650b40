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
  c4        configs[3]: IVF nlist=4096 nprobe=32 over 10M x 128: IVFIndex.train on 256 x 4096 rows (k-means++ and 20
            Lloyd steps, seed 42), assignment of all rows, inverted lists spread over the ranks by whole lists
            (ShardedIVF), batch 1024 and single queries, the scan kernel against the HBM roofline, a sample of the
            queries against the oracle and the oracle timed on the host cores;
  c5        configs[4]: PQ m=16 nbits=8 ADC scan over 100M codes (rows / N per GPU, HipPqBackend), batch 1 and 8, no
            filter / 50 % / 1 % bitmask resident in HBM, the scan kernel against the HBM roofline, the oracle on a
            sample of the codes; the PQIndex class with a filter callable (first and repeated search); Manhattan
            1M x 128 batch 1 and 1024 beside it;
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


def note(msg: str) -> None:
    """Progress on stderr (rank 0): the JSON line on stdout stays the only stdout output."""
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


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
    if visible < n and os.environ.get("VDB_BENCH_BACKEND", "nccl") == "nccl":
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
    ap.add_argument("--no-extra", action="store_true", help="skip the side measurements (prefilter, c3, c4, c5, c1, ceilings)")
    ap.add_argument("--skip", default="", help="side measurements to leave out, e.g. c3,c5")
    ap.add_argument("--side-scale", type=float, default=1.0, help="scale the row counts of c4 and c5 (rehearsals only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    import torch.distributed as dist
    # VDB_BENCH_BACKEND=gloo rehearses the N-rank code on fewer GPUs than ranks (ranks share devices, the gathers go
    # through the host): a logic check, not a measurement
    dist_backend = os.environ.get("VDB_BENCH_BACKEND", "nccl")
    if dist_backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=dist_backend)
            gather = dist.all_gather_into_tensor

            def gather_through_host(out, inp, group=None):          # gloo does not gather device tensors
                staged = torch.empty(out.shape, dtype=out.dtype)
                gather(staged, inp.cpu(), group=group)
                out.copy_(staged)
            dist.all_gather_into_tensor = gather_through_host

    from vectordb_amd import _native
    from vectordb_amd.sharded import HipFlatBackend, ShardedFlat, shard_range
    _native.init(local_rank)

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x: float) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], device=dev if dist_backend == "nccl" else "cpu", dtype=torch.float64)
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

    note(f"rows resident ({hi - lo} on this GPU), timing the headline")
    elapsed, kernel_ms, launches, (out_d, out_r) = timed_steps(index, queries, k, args.steps, args.warmup, barrier, _native)
    note("headline timed")
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
        skip = set(filter(None, args.skip.split(",")))
        if "c3" not in skip:
            note("side measurement c3")
            try:
                extra["c3"] = bench_c3(world, rank, dev, gen, barrier, max_over_ranks, _native)
            except Exception as e:          # a side measurement must not take the headline line with it
                extra["c3"] = {"error": f"{type(e).__name__}: {e}"}
        for name, fn in (("c4", bench_c4), ("c5", bench_c5)):
            if name in skip:
                continue
            note(f"side measurement {name}")
            try:
                extra[name] = fn(world, rank, dev, gen, barrier, max_over_ranks, _native, scale=args.side_scale)
            except Exception as e:
                extra[name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
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
            **({"rehearsal": f"{dist_backend} process group, ranks sharing GPUs: not a measurement"} if dist_backend != "nccl" else {}),
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


C4_ROWS, C4_DIM, C4_LISTS, C4_PROBE = 10_000_000, 128, 4096, 32
C5_ROWS, C5_M = 100_000_000, 16


def hbm_roofline(nbytes: float, kernel_s: float, **more) -> dict:
    gbs = nbytes / kernel_s / 1e9 if kernel_s and kernel_s == kernel_s and kernel_s > 0 else 0.0
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes_per_launch": int(nbytes), **more}


class _Probed:
    """ShardedIVF behind the (queries, k) call that timed_steps makes."""

    def __init__(self, index, n_probe):
        self.index, self.n_probe = index, n_probe

    def search(self, queries, k):
        return self.index.search(queries, k, self.n_probe)


class _Masked:
    """ShardedFlat with this rank's filter words bound."""

    def __init__(self, index, words):
        self.index, self.words = index, words

    def search(self, queries, k):
        return self.index.search(queries, k, self.words)


def bench_c4(world, rank, dev, gen, barrier, max_over_ranks, native, cpu_sample=16, scale=1.0):
    """BASELINE.json configs[3]: IVF nlist=4096 nprobe=32 over 10M x 128 standard-normal rows, k = 10.  Training is the
    reference's: k-means++ and 20 Lloyd steps over 256 rows per cluster (index/ivf.py:126-231 through IVFIndex.train),
    then every row goes to its nearest centroid (ivf.py:553-565).  Whole lists are dealt to the ranks by length; a step
    is: rank all centroids, scan the probed lists this rank owns [-> all-gather -> merge by (distance, visiting order)]."""
    from oracle import ref_numpy as R
    from vectordb_amd.index import IVFIndex
    from vectordb_amd.sharded import HipIvfBackend, ShardedIVF, assign_lists_balanced, local_ivf_layout
    n, d, nlist, nprobe, k = int(C4_ROWS * scale), C4_DIM, max(int(C4_LISTS * scale), 64), C4_PROBE, TOP_K
    x = np.empty((n, d), np.float32)
    for c0 in range(0, n, 16 * GEN_BLOCK):
        c1 = min(n, c0 + 16 * GEN_BLOCK)
        x[c0:c1] = block_rows(c0, c1, d, 777, dev, gen).cpu().numpy()
    ix = IVFIndex(dimension=d, metric="euclidean", n_clusters=nlist, n_probe=nprobe, n_iter=20, seed=42)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix.train(x[:256 * nlist])
    train_s = time.perf_counter() - t0
    note(f"c4: trained {nlist} centroids in {train_s:.2f} s")
    ph = dict(ix._last_kmeans.phase_seconds)
    cent = ix.centroids
    t0 = time.perf_counter()
    assign, _ = native.assign(x, cent, native.EUCLIDEAN)
    assign_s = time.perf_counter() - t0
    lengths = np.bincount(assign, minlength=nlist)
    owner = assign_lists_balanced(lengths, world)
    t0 = time.perf_counter()
    mine, list_rows_local, offsets = local_ivf_layout(assign, nlist, owner, rank)
    backend = HipIvfBackend(d, native.EUCLIDEAN, cent, x[mine], list_rows_local, offsets, mine)
    build_s = time.perf_counter() - t0
    note(f"c4: lists built ({len(mine)} rows on this GPU)")
    index = _Probed(ShardedIVF(backend, rank, world), nprobe)
    out = {"workload": f"IVF nlist={nlist} nprobe={nprobe}, {n} x {d} fp32, k={k}, whole lists dealt to {world} GPU(s) (BASELINE configs[3])",
           "train": {"what": f"IVFIndex.train: k-means++ + 20 Lloyd steps over {256 * nlist} rows, {nlist} clusters, seed 42",
                     "seconds": round(train_s, 3), "init_pp_s": round(ph.get("init_pp", 0.0), 3), "lloyd_s": round(ph.get("lloyd", 0.0), 3),
                     "lloyd_steps": ph.get("lloyd_steps")},
           "assign_all_rows_s": round(assign_s, 3), "build_lists_s": round(build_s, 3),
           "rows_on_this_gpu": int(len(mine)), "longest_list": int(lengths.max())}
    gen.manual_seed(999)
    queries = torch.randn((N_QUERIES, d), device=dev, dtype=torch.float32, generator=gen)
    results = None
    for nq, steps, warmup in ((N_QUERIES, 10, 3), (1, 50, 5)):
        q = queries[:nq].contiguous()
        elapsed, kernel_ms, launches, (od, orow) = timed_steps(index, q, k, steps, warmup, barrier, native)
        elapsed = max_over_ranks(elapsed)
        used_mfma, overflow = backend.lists.search_stats()
        scanned = backend.lists.last_scan_rows()                     # (row, query) pairs this rank scanned in one step
        per_step_ms = kernel_ms / steps
        if used_mfma:
            # the timed kernel is the matrix-core pass over the bf16 copy.  A chunk of a list is read once for all the
            # queries of the batch that probe it, so what HBM must deliver is every probed list once: at most this
            # GPU's whole copy (2 bytes per element), and not more than one read per scanned (row, query) pair
            roof = hbm_roofline(min(scanned, len(mine)) * d * 2.0, per_step_ms * 1e-3,
                                kernel="ivf_approx_kernel (bf16 copy of the member rows)",
                                kernel_ms=round(per_step_ms, 4), launches_per_step=launches // steps,
                                per_query_fp32_bytes_times_queries=int(scanned * d * 4 + nq * nlist * d * 4))
        else:
            roof = hbm_roofline(scanned * d * 4.0 + nq * nlist * d * 4.0, elapsed / steps,
                                kernel="centroid ranking + list scan + selection (whole step)", kernel_ms_timed=round(per_step_ms, 4),
                                launches_per_step=launches // steps)
        out[f"batch_{nq}"] = {"queries_per_s": round(nq * steps / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 4),
                              "row_query_pairs_scanned_on_this_gpu": int(scanned), "matrix_core_scan": used_mfma,
                              "queue_overflow": overflow, "roofline": roof}
        if nq == N_QUERIES:
            results = (od.cpu().numpy(), orow.cpu().numpy())
    # parity sample and the host baseline: the oracle's IVFIndex.search over all 10M rows, one query at a time
    if rank == 0 and cpu_sample > 0:
        order = np.argsort(assign, kind="stable")
        offs = np.searchsorted(assign[order], np.arange(nlist + 1))
        members = [order[offs[l]:offs[l + 1]] for l in range(nlist)]
        qh = queries[:cpu_sample].cpu().numpy()
        R.ivf_search(qh[0], x, cent, members, k, nprobe, "euclidean")
        t0 = time.perf_counter()
        ref = [R.ivf_search(qh[i], x, cent, members, k, nprobe, "euclidean") for i in range(cpu_sample)]
        cpu_s = time.perf_counter() - t0
        same = all(np.array_equal(results[1][i], ref[i][0]) and np.array_equal(results[0][i], ref[i][1].astype(np.float32))
                   for i in range(cpu_sample))
        out["identical_to_oracle"] = bool(same)
        if world == 1:
            out["cpu_baseline"] = {"value": round(cpu_sample / cpu_s, 2), "unit": "queries/s", "cores": host_threads(), "kind": "port",
                                   "sample": f"{cpu_sample} of the queries, one at a time, over all {n} rows (vectorised NumPy restatement "
                                             "of IVFIndex.search; the reference itself loops over the members in Python)"}
    backend.lists.close()
    return out


def pq_block(lo: int, hi: int, m: int, seed: int, dev, gen, block: int = 1_000_000) -> "torch.Tensor":
    """Codes of global rows [lo, hi): uniform bytes, block b of `block` rows from seed + b (the same at every world size)."""
    parts = []
    for b in range(lo // block, (hi + block - 1) // block):
        gen.manual_seed(seed + b)
        blk = torch.randint(0, 256, (block, m), device=dev, dtype=torch.uint8, generator=gen)
        b0 = b * block
        parts.append(blk[max(lo - b0, 0):min(hi - b0, block)])
    return (parts[0] if len(parts) == 1 else torch.cat(parts)).contiguous()


def keep_bits(lo: int, hi: int, fraction: float, seed: int, dev, gen, block: int = 1_000_000) -> "torch.Tensor":
    """Filter outcome (bool) of global rows [lo, hi): row kept with probability `fraction`, block-seeded like the codes."""
    parts = []
    for b in range(lo // block, (hi + block - 1) // block):
        gen.manual_seed(seed + b)
        blk = torch.rand(block, device=dev, generator=gen) < fraction
        b0 = b * block
        parts.append(blk[max(lo - b0, 0):min(hi - b0, block)])
    return parts[0] if len(parts) == 1 else torch.cat(parts)


def pack_bits(bits: "torch.Tensor") -> "torch.Tensor":
    """bool [n] -> int32 words, bit i of word i // 32 = bits[i] (the layout every *_search entry point takes)."""
    n = bits.numel()
    padded = torch.zeros(((n + 31) // 32) * 32, device=bits.device, dtype=torch.int64)
    padded[:n] = bits
    weights = 2 ** torch.arange(32, device=bits.device, dtype=torch.int64)
    return (padded.view(-1, 32) * weights).sum(1).to(torch.int32).contiguous()          # wraps to the same 32 bits


def bench_c5(world, rank, dev, gen, barrier, max_over_ranks, native, cpu_codes=2_000_000, cpu_queries=4, scale=1.0):
    """BASELINE.json configs[4]: PQ m=16 nbits=8 ADC scan over 100M codes (1.6 GB) with a filter bitmask, k = 10,
    rows / world per GPU; a step is: lookup tables -> scan of this rank's codes under its slice of the bitmask ->
    selection [-> all-gather -> merge].  Beside it the PQIndex class with a filter callable, and Manhattan 1M x 128."""
    from oracle import ref_numpy as R
    from vectordb_amd.sharded import HipFlatBackend, HipPqBackend, ShardedFlat, shard_range
    n, m, d, k = int(C5_ROWS * scale), C5_M, C4_DIM, TOP_K
    cpu_codes = min(cpu_codes, n)
    lo, hi = shard_range(n, world, rank)
    book = np.random.default_rng(7).standard_normal((m, 256, d // m), dtype=np.float32)
    backend = HipPqBackend(d, m, 256, book)
    for c0 in range(lo, hi, 10_000_000):
        backend.add_device(pq_block(c0, min(hi, c0 + 10_000_000), m, 5000, dev, gen))
        torch.cuda.synchronize()
    index = ShardedFlat(backend, n, rank, world)
    note(f"c5: {hi - lo} codes resident")
    gen.manual_seed(555)
    queries = torch.randn((8, d), device=dev, dtype=torch.float32, generator=gen)
    out = {"workload": f"PQ m={m} nbits=8 ADC scan, {n} codes, k={k}, rows/{world} per GPU (BASELINE configs[4])",
           "codes_on_this_gpu": hi - lo, "scans": []}
    checks = {}
    for sel in (None, 0.5, 0.01):
        words = None
        if sel is not None:
            words = pack_bits(keep_bits(lo, hi, sel, 9000, dev, gen))
        for nq in (1, 8):
            q = queries[:nq].contiguous()
            # (a scan of 0.4 ms: the clocks settle over the first ~70 back-to-back launches, 0.40 -> 0.32 ms for the
            # byte-table kernel, so the warm-up is as long as that and says so)
            c5_steps, c5_warmup = 50, 100
            elapsed, kernel_ms, launches, (od, orow) = timed_steps(_Masked(index, words), q, k, c5_steps, c5_warmup, barrier, native)
            elapsed = max_over_ranks(elapsed)
            kern = kernel_ms / max(launches, 1)
            byte_tables, gave_up = backend.codes.search_stats()
            # queries that share one pass over the codes: eight on the one-byte tables, four on the fp32 tables
            passes = (nq + 7) // 8 if byte_tables else (nq + 3) // 4
            nbytes = passes * ((hi - lo) * m + ((hi - lo) // 8 if words is not None else 0))
            out["scans"].append({"batch": nq, "filter": sel, "queries_per_s": round(nq * c5_steps / elapsed, 2),
                                 "ms_per_step": round(elapsed / c5_steps * 1e3, 4), "steps": c5_steps, "warmup": c5_warmup,
                                 "passes_over_the_codes": passes,
                                 "byte_tables": byte_tables, "fp32_scan_repeated_the_batch": gave_up,
                                 "roofline": hbm_roofline(nbytes, kern * 1e-3, kernel="pq_scan8_kernel" if byte_tables else "pq_scan_kernel", kernel_ms=round(kern, 4))})
            if nq == 8:
                checks[sel] = (od.cpu().numpy(), orow.cpu().numpy())
        del words
    # the oracle on the first cpu_codes rows: the device result restricted to them must be a prefix-compatible subset, so the
    # check searches a second store holding just those rows (rank 0, one GPU's worth of work, not timed)
    if rank == 0:
        small = HipPqBackend(d, m, 256, book)
        codes_s = pq_block(0, cpu_codes, m, 5000, dev, gen)
        small.add_device(codes_s)
        codes_h = codes_s.cpu().numpy()
        keep_h = keep_bits(0, cpu_codes, 0.01, 9000, dev, gen)
        words_s = pack_bits(keep_h)
        keep_h = keep_h.cpu().numpy()
        qh = queries[:cpu_queries].cpu().numpy()
        sd = torch.empty((cpu_queries, k), device=dev, dtype=torch.float32)
        sr = torch.empty((cpu_queries, k), device=dev, dtype=torch.int64)
        ok = True
        t_cpu = 0.0
        for mask_h, words in ((None, None), (keep_h, words_s)):
            small.local_search(queries[:cpu_queries].contiguous(), k, sd, sr, words)
            torch.cuda.synchronize()
            for i in range(cpu_queries):
                t0 = time.perf_counter()
                rows_ref, dist_ref = R.pq_search(qh[i], book, codes_h, k, mask_h)
                if mask_h is None:
                    t_cpu += time.perf_counter() - t0
                ok &= bool(np.array_equal(sr[i].cpu().numpy(), rows_ref)) and bool(np.array_equal(sd[i].cpu().numpy(), dist_ref))
        out["identical_to_oracle_on_sample"] = bool(ok)
        if world == 1:
            out["cpu_baseline"] = {"value": round(cpu_queries / t_cpu * cpu_codes / n, 4), "unit": "queries/s", "cores": host_threads(),
                                   "kind": "port", "measured_queries_per_s_on_sample": round(cpu_queries / t_cpu, 3),
                                   "sample": f"{cpu_queries} single queries over the first {cpu_codes} of the {n} codes, no filter, scaled "
                                             "linearly in the code count (vectorised NumPy restatement of PQIndex.search; the reference "
                                             "itself scans in a Python loop)"}
        small.codes.close()
        del small, codes_s, words_s
    backend.codes.close()
    del index, backend
    torch.cuda.empty_cache()

    if world == 1:
        note("c5: PQIndex class with a filter callable")
        out["pq_index_class"] = pq_class_filtered(native)
    note("c5: Manhattan")

    # Manhattan side-check: Flat 1M x 128, rows / world per GPU, batch 1 and 1024
    n1, d1 = N_ROWS, C4_DIM
    lo, hi = shard_range(n1, world, rank)
    flat = HipFlatBackend(d1, native.MANHATTAN, prefilter=None)
    fill_store(flat, lo, hi, d1, 808, dev, gen)
    findex = ShardedFlat(flat, n1, rank, world)
    gen.manual_seed(556)
    q_all = torch.randn((N_QUERIES, d1), device=dev, dtype=torch.float32, generator=gen)
    man = {"workload": f"Flat manhattan {n1} x {d1} fp32, k={k}, rows/{world} per GPU"}
    for nq, steps in ((1, 20), (N_QUERIES, 5)):
        q = q_all[:nq].contiguous()
        elapsed, kernel_ms, launches, (od, orow) = timed_steps(findex, q, k, steps, 2, barrier, native)
        elapsed = max_over_ranks(elapsed)
        kern = kernel_ms / max(launches, 1)
        entry = {"queries_per_s": round(nq * steps / elapsed, 1), "ms_per_step": round(elapsed / steps * 1e3, 4),
                 "kernel_ms_avg": round(kern, 4), "kernels_per_step": launches // steps}
        if nq == 1:
            entry["roofline"] = hbm_roofline((hi - lo) * d1 * 4.0, kern * 1e-3, kernel="streaming row scan")
        else:
            entry["valu_tops"] = round(3.0 * nq * (hi - lo) * d1 * steps / elapsed / 1e12, 2)     # subtract, abs, add per element pair
        if rank == 0 and nq == 1:
            rows_h = np.empty((n1, d1), np.float32)
            for c0 in range(0, n1, 16 * GEN_BLOCK):
                rows_h[c0:c0 + 16 * GEN_BLOCK] = block_rows(c0, min(n1, c0 + 16 * GEN_BLOCK), d1, 808, dev, gen).cpu().numpy()
            qh = q_all[:4].cpu().numpy()
            t0 = time.perf_counter()
            ref = [R.flat_search(qh[i], rows_h, k, "manhattan") for i in range(4)]
            cpu_s = time.perf_counter() - t0
            entry["identical_rows_to_oracle"] = bool(np.array_equal(orow[0].cpu().numpy(), ref[0][0]))
            entry["distances_ok"] = bool(np.allclose(od[0].cpu().numpy(), ref[0][1], rtol=1e-4, atol=1e-4))
            if world == 1:
                entry["cpu_queries_per_s"] = round(4 / cpu_s, 2)
            del rows_h
        man[f"batch_{nq}"] = entry
    out["manhattan"] = man
    flat.store.close()
    return out


def pq_class_filtered(native, n=1_000_000, d=C4_DIM, m=C5_M):
    """PQIndex.search with a filter callable, as a user of the reference's class calls it (index/pq.py:247-318), on 1M rows:
    the first filtered search runs the callable over every row and uploads the packed words; repeats of the same filter
    find the words in HBM (the callable is not called again until a write)."""
    from vectordb_amd.index import PQIndex
    rng = np.random.default_rng(31)
    x = rng.standard_normal((n, d), dtype=np.float32)
    ix = PQIndex(dimension=d, n_subvectors=m, n_bits=8, n_iter=5, seed=1)
    ix.train(x[:20_000])
    t0 = time.perf_counter()
    ix.add_batch([f"r{i}" for i in range(n)], x, [{"cat": i % 100} for i in range(n)])
    add_s = time.perf_counter() - t0
    calls = [0]

    def only_cat_7(_id, meta):
        calls[0] += 1
        return meta["cat"] == 7

    q = rng.standard_normal(d, dtype=np.float32)
    ix.search(q, 10)
    t0 = time.perf_counter()
    first = ix.search(q, 10, filter_fn=only_cat_7)
    first_ms = (time.perf_counter() - t0) * 1e3
    calls_first = calls[0]
    t0 = time.perf_counter()
    for _ in range(20):
        again = ix.search(q, 10, filter_fn=only_cat_7)
    repeat_ms = (time.perf_counter() - t0) * 1e3 / 20
    t0 = time.perf_counter()
    for _ in range(20):
        ix.search(q, 10)
    plain_ms = (time.perf_counter() - t0) * 1e3 / 20
    return {"what": f"PQIndex.search(filter_fn=...) over {n} rows, 1 % of the rows pass", "add_batch_s": round(add_s, 2),
            "first_filtered_search_ms": round(first_ms, 2), "filter_calls_first": calls_first,
            "repeated_filtered_search_ms": round(repeat_ms, 3), "filter_calls_in_20_repeats": calls[0] - calls_first,
            "unfiltered_search_ms": round(plain_ms, 3),
            "same_results": [r.id for r in first] == [r.id for r in again]}


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
