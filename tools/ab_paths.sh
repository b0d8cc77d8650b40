#!/bin/bash
# A/B timing of library builds under vectordb_amd/_lib/exp/ on bench_paths.py items:
#   gpurun -- 'bash tools/ab_paths.sh bf16'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for lib in vectordb_amd/_lib/exp/*.so; do
  echo "== $(basename $lib .so)"
  VDB_LIB_PATH=$R/$lib timeout -k 10 200 python bench_paths.py --only "$1" 2>&1 | grep -o "\"workload\": \"[^\"]*\"\|\"kernel_ms\": [0-9.]*\|\"ms_per_batch\": [0-9.]*" | paste - - -
done
