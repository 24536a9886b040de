set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
timeout 800 python scripts/stats_probe.py 256 2>&1 | grep -A 30 "checksum=False" | tail -28
for B in 256 2560; do
  timeout 900 python bench.py --blocks $B --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench_$B.err | tee gpurun_out/bench_$B.json
  tail -2 gpurun_out/bench_$B.err
done
