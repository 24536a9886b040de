set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc; lscpu | grep -E "Model name|Socket|Thread|Core" ; free -g | head -2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
for B in 256 2560 6144; do
  timeout 900 python bench.py --blocks $B --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench_$B.err | tee gpurun_out/bench_$B.json
  tail -2 gpurun_out/bench_$B.err
done
