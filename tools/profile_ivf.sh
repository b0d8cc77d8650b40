#!/bin/bash
# rocprofv3 counters of the IVF batch scan (bench_paths.py --only ivf):
#   gpurun -- 'bash tools/profile_ivf.sh'
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_ivfpmc
mkdir -p "$OUT"
cd "$R"
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/sq1" -- python3 bench_paths.py --only ivf > "$OUT/sq1.log" 2>&1 || exit 1
timeout -s KILL 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM --output-format csv -d "$OUT/sq2" -- python3 bench_paths.py --only ivf > "$OUT/sq2.log" 2>&1 || exit 1
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 bench_paths.py --only ivf > "$OUT/fetch.log" 2>&1 || exit 1
echo done
