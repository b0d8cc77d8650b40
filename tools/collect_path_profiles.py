#!/usr/bin/env python3
"""Copy what tools/pmc_paths.sh left under gpurun_out/pmc_<tag>/ into profiles/:
    python tools/collect_path_profiles.py <tag> <name>
writes profiles/<name>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, kernel names cut to 110
characters) and profiles/<name>_pmc.csv (per kernel and counter: dispatches, mean and maximum per dispatch; the
maximum is the benchmarked workload where a kernel also runs on small inputs).  HBM bytes of a dispatch:
2 * FETCH_SIZE KiB + WRITE_SIZE KiB (FETCH_SIZE counts half of a wide coalesced read on gfx950)."""
import collections, csv, glob, os, sys

tag, name = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "pmc_" + tag)
stats = glob.glob(src + "/stats/*/*kernel_stats.csv")
rows = list(csv.reader(open(stats[0])))
with open(os.path.join(root, "profiles", name + "_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        if r and (r[0] == "Name" or r[0].startswith("void vdb::") or r[0].startswith("vdb::")):
            r[0] = r[0][:110]
            w.writerow(r)
with open(os.path.join(root, "profiles", name + "_pmc.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["pass", "kernel", "counter", "dispatches", "mean", "max"])
    for p in sorted(glob.glob(src + "/pmc_*/*/*counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(p)):
            acc[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        pass_name = os.path.relpath(p, src).split(os.sep)[0]
        for k, cs in sorted(acc.items()):
            if "vdb::" not in k:
                continue
            for c, v in sorted(cs.items()):
                w.writerow([pass_name, k, c, len(v), f"{sum(v) / len(v):.6g}", f"{max(v):.6g}"])
print("wrote profiles/%s_kernel_stats.csv and profiles/%s_pmc.csv" % (name, name))
