set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1200 python bench.py 2>gpurun_out/bench_default.err | tee gpurun_out/bench_default.json
tail -3 gpurun_out/bench_default.err
timeout 600 python bench.py --blocks 8192 --steps 2 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench_8192.err | tee gpurun_out/bench_8192.json
