#!/usr/bin/env python3
"""Per-kernel HBM traffic of the streaming paths from the passes of tools/pmc_paths.sh (+ a WRITE_SIZE pass):
    python tools/collect_paths_pmc.py r01d
Dispatches are matched across passes by their order per kernel name (the run is deterministic); only the
long dispatches of a kernel (at least half its longest) are kept, i.e. the benchmarked workloads.
HBM bytes = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB (the gfx950 reading of the two counters).
Writes profiles/<tag>_paths_pmc.csv."""
import csv, glob, os, sys, collections
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "pmc_" + tag)


def per_kernel(path, value):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        v = value(r)
        if v is not None:
            out[r["Kernel_Name"]].append(v)
    return out


trace = sorted(csv.DictReader(open(glob.glob(src + "/stats/*/*kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
dur = collections.defaultdict(list)
for r in trace:
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
fetch = per_kernel(glob.glob(src + "/pmc_fetch/*/*counter_collection.csv")[0], lambda r: float(r["Counter_Value"]) if r["Counter_Name"] == "FETCH_SIZE" else None)
write = per_kernel(glob.glob(src + "/pmc_write/*/*counter_collection.csv")[0], lambda r: float(r["Counter_Value"]) if r["Counter_Name"] == "WRITE_SIZE" else None)
rows = []
for k, ds in dur.items():
    if not any(s in k for s in ("stream_sums_kernel", "pq_scan_kernel", "pq_scan8_kernel", "topk_cand_kernel", "ivf_approx_kernel",
                                "ivf_scatter_kernel", "ivf_finish_kernel", "ivf_cent_keys_kernel", "ivf_bound_kernel",
                                "pq_exact8_kernel", "rowreg_sums_kernel")):
        continue
    if len(fetch.get(k, [])) != len(ds) or len(write.get(k, [])) != len(ds):
        continue
    top = max(ds)
    idx = [i for i, d in enumerate(ds) if d >= 0.5 * top]
    d = sum(ds[i] for i in idx) / len(idx)
    f = sum(fetch[k][i] for i in idx) / len(idx)
    w = sum(write[k][i] for i in idx) / len(idx)
    hbm = (2 * f + w) * 1024
    rows.append((k, len(idx), round(d / 1e3, 1), round(f, 1), round(w, 1), round(hbm / 1e6, 1), round(hbm / d, 1)))
out = os.path.join(root, "profiles", tag + "_paths_pmc.csv")
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "dispatches", "mean_us", "fetch_size_kib", "write_size_kib", "hbm_mb", "hbm_gb_per_s"])
    w.writerows(rows)
for r in rows:
    print(r)
