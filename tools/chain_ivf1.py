#!/usr/bin/env python3
"""Kernel chain of one single-query IVF search from a rocprofv3 kernel trace of `bench_paths.py --only ivf`:
    python tools/chain_ivf1.py gpurun_out/prof_ivf1/ivf_kernel_trace.csv"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"][:72], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"])) for r in rows]
idx = [i for i, n in enumerate(names) if "ivf_one_plan_kernel" in n[0]]
i = idx[len(idx) // 2]
t0 = names[i - 1][2]
for n in names[i - 1:i + 4]:
    print(f"{(n[2] - t0) / 1e3:8.1f} us  +{n[1] / 1e3:6.1f} us  {n[0]}")
