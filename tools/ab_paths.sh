#!/bin/bash
# A/B timing of library builds under vectordb_amd/_lib/exp/ on bench_paths.py paths, one GPU box for all:
#   gpurun -- 'bash tools/ab_paths.sh ivf,l1batch'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for lib in vectordb_amd/_lib/exp/*.so; do
  echo "== $(basename $lib .so)"
  VDB_LIB_PATH=$R/$lib timeout -k 10 200 python bench_paths.py --only "${1:-ivf}" 2>&1 | grep -o "\"workload\": \"[^\"]*\", \"qps\": [0-9.]*, \"ms_per_batch\": [0-9.]*" || exit 1
done
