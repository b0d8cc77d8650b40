#!/bin/bash
# bench.py's JSON line and every bench_paths.py line of one build, written under gpurun_out/final/ on the GPU box:
#   gpurun --timeout 1200 -- 'bash tools/final_evidence.sh'
# Copy gpurun_out/final/bench.json (last line) and paths.jsonl into profiles/ under the build's tag afterwards.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 600 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || { echo bench failed; tail -n 5 gpurun_out/final/bench.err; exit 1; }
echo "bench done"; tail -c 600 gpurun_out/final/bench.json
timeout -k 10 900 python bench_paths.py > gpurun_out/final/paths.jsonl 2> gpurun_out/final/paths.err || { echo paths failed; tail -n 5 gpurun_out/final/paths.err; exit 1; }
echo "paths done"
