#!/bin/bash
# rocprofv3 evidence of one build, summarised on the GPU box (the raw traces exceed what gpurun copies back):
#   gpurun --timeout 1200 -- 'bash tools/profile_all.sh r02c "bench pq bf16 assign,l1batch ivf"'
# For every path: tools/pmc_paths.sh (kernel stats + counter passes), then the summaries that
# tools/collect_path_profiles.py and tools/collect_paths_pmc.py write, copied to gpurun_out/summ/; the raw
# output is deleted.  Copy gpurun_out/summ/* into profiles/ afterwards.
set -u
TAG=${1:-r02}
PATHS=${2:-"bench pq bf16 assign,l1batch ivf"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
mkdir -p gpurun_out/summ
for P in $PATHS; do
    NAME=${P//,/_}
    if [ "$P" = bench ]; then
        bash tools/profile_bench.sh "$TAG" || { echo "profile_bench failed"; tail -5 gpurun_out/prof_$TAG/*.log; exit 1; }
        python3 tools/collect_profiles.py "$TAG" || exit 1
        cp profiles/${TAG}_bench_kernel_stats.csv profiles/${TAG}_bench_pmc.csv profiles/bench_traffic.json gpurun_out/summ/
        rm -rf gpurun_out/prof_$TAG
    else
        SCALE=1.0
        bash tools/pmc_paths.sh "$P" $SCALE "${TAG}_$NAME" || { echo "pmc_paths $P failed"; tail -5 gpurun_out/pmc_${TAG}_$NAME/*.log; exit 1; }
        python3 tools/collect_path_profiles.py "${TAG}_$NAME" "${TAG}_$NAME" || exit 1
        python3 tools/collect_paths_pmc.py "${TAG}_$NAME" || exit 1
        cp profiles/${TAG}_${NAME}_kernel_stats.csv profiles/${TAG}_${NAME}_pmc.csv profiles/${TAG}_${NAME}_paths_pmc.csv gpurun_out/summ/
        rm -rf gpurun_out/pmc_${TAG}_$NAME
    fi
    echo "== $P done"
done
