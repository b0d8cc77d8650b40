#!/bin/bash
# rocprofv3 counter passes for one path of bench_paths.py.  Run through gpurun from the repo root:
#   gpurun -- 'bash tools/pmc_paths.sh ivf 0.2 tag'
set -u
ONLY=${1:-ivf}
SCALE=${2:-0.2}
TAG=${3:-paths}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
ARGS="bench_paths.py --only $ONLY --scale $SCALE"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS > "$OUT/stats.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq1" -- python3 $ARGS > "$OUT/pmc_sq1.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/pmc_sq2" -- python3 $ARGS > "$OUT/pmc_sq2.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1 || exit 1
echo done
