#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of gpurun_out/prof_<tag>/ into profiles/ (kernel names shortened)."""
import collections, csv, glob, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)

stats = glob.glob(f"{src}/stats/*/*_kernel_stats.csv")
if stats:
    rows = list(csv.reader(open(stats[0])))
    with open(f"profiles/{tag}_bench_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:100]
            w.writerow(r)

pmc_raw = sorted(glob.glob(f"{src}/pmc_*/*/*_counter_collection.csv"))
if not pmc_raw:
    sys.exit(f"no raw counter output under {src}: nothing rewritten")
with open(f"profiles/{tag}_bench_pmc.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["pass", "kernel", "counter", "dispatches", "mean_per_dispatch"])
    for p in pmc_raw:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            acc[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        name = p.split("/")[2]
        for k, cs in acc.items():
            if not k.startswith("void vdb::"):
                continue
            for c, v in cs.items():
                w.writerow([name, k, c, len(v), f"{sum(v) / len(v):.6g}"])

# HBM traffic of the dominant kernel per launch, corrected as the gfx950 notes prescribe:
# FETCH_SIZE (KiB) under-counts wide coalesced reads by 2x; WRITE_SIZE (KiB) is exact.
import json
vals = {}
for r in csv.DictReader(open(f"profiles/{tag}_bench_pmc.csv")):
    # the headline's full pass: metric 2 (cosine), fp32 rows, last template argument (mode) 0 = the queue pass; the
    # side measurements of bench.py run the same kernel for other metrics and sizes
    if "flat_queue_fast_kernel<2, false, 0>" in r["kernel"] and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
        vals[r["counter"]] = float(r["mean_per_dispatch"])
if len(vals) == 2:
    traffic = 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024
    json.dump({"kernel": "flat_queue_fast_kernel<cosine> (full pass)", "tag": tag, "fetch_size_kib": vals["FETCH_SIZE"],
               "write_size_kib": vals["WRITE_SIZE"], "hbm_bytes_per_launch": traffic,
               "note": "2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes of `bench.py --steps 3 --warmup 1`"},
              open("profiles/bench_traffic.json", "w"), indent=1)
    print("traffic bytes/launch", traffic)

