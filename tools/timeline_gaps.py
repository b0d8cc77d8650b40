#!/usr/bin/env python3
"""Kernel timeline of the last search in a rocprofv3 --kernel-trace run: start offset, duration and the gap
before every kernel (us).  usage: timeline_gaps.py <rocprof output dir> [first-kernel-substring] [count]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
tail = rows[-n:]
t0 = int(tail[0]['Start_Timestamp']); prev = t0
busy = 0
for r in tail:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"at {(s-t0)/1e3:8.1f}  gap {(s-prev)/1e3:6.1f}  dur {(e-s)/1e3:7.1f}  {r['Kernel_Name'][:70]}")
    busy += e - s; prev = e
print(f"span {(prev-t0)/1e3:.1f} us, busy {busy/1e3:.1f} us")
