set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 600 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err || { echo bench failed; tail -n 5 gpurun_out/final/bench.err; exit 1; }
echo "bench done"; tail -c 600 gpurun_out/final/bench.json
timeout -k 10 900 python bench_paths.py > gpurun_out/final/paths.jsonl 2> gpurun_out/final/paths.err || { echo paths failed; tail -n 5 gpurun_out/final/paths.err; exit 1; }
echo "paths done"
