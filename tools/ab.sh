#!/bin/bash
# A/B timing of library builds under vectordb_amd/_lib/exp/ on one GPU box (same device for all):
#   gpurun -- 'bash tools/ab.sh [bench args]'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for rep in 1 2; do
  for lib in vectordb_amd/_lib/exp/*.so; do
    printf "%s " "$(basename $lib .so)"
    VDB_LIB_PATH=$R/$lib timeout -k 10 100 python bench.py --cpu-sample 0 --steps 20 "$@" 2>&1 | grep -o "\"ms_per_step\": [0-9.]*\|\"kernel_ms\": [0-9.]*\|\"parity[a-z_]*\": [a-z]*" | tr "\n" " "
    echo
  done
done
