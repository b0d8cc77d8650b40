#!/bin/bash
# FETCH_SIZE of the IVF list scan for library builds under vectordb_amd/_lib/exp/ (and group widths)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for lib in vectordb_amd/_lib/exp/*.so; do
  for G in 8 16; do
    OUT=gpurun_out/fetch_$(basename $lib .so)_$G
    rm -rf $OUT
    VDB_LIB_PATH=$R/$lib VDB_STREAM_GROUP=$G timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 bench_paths.py --only ivf > /dev/null 2>&1 || exit 1
    python3 - "$OUT" "$(basename $lib .so) G=$G" <<'P'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(p)):
        if "stream_sums_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][10:48]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(sys.argv[2], k, len(v), "max FETCH KiB %.0f -> %.2f GB" % (max(v), 2 * max(v) * 1024 / 1e9))
P
    rm -rf $OUT
  done
done
