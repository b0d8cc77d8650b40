#!/bin/bash
# rocprofv3 evidence for bench.py on the GPU box.  Run through gpurun from the repo root:
#   gpurun -- 'bash tools/profile_bench.sh r01'
# Writes raw output under gpurun_out/prof_<tag>/; copy the summaries into profiles/ afterwards
# (python tools/collect_profiles.py <tag>).
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd "$R"
ARGS="bench.py --steps 20 --warmup 3 --cpu-sample 0"       # bench.py's default step counts
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS > "$OUT/stats.log" 2>&1 || exit 1
PMCARGS="bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extra"      # the headline alone: counters serialise the kernels
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq1" -- python3 $PMCARGS > "$OUT/pmc_sq1.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq2" -- python3 $PMCARGS > "$OUT/pmc_sq2.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $PMCARGS > "$OUT/pmc_fetch.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $PMCARGS > "$OUT/pmc_write.log" 2>&1 || exit 1
echo done
