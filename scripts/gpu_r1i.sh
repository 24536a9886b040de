set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -4
timeout 800 python scripts/stats_probe.py 256 2>&1 | grep -A 24 "checksum=False" | egrep "grid|twinstop|sat|longback|seq_grid|walkiter|per grid|MiB/s"
for B in 2560 6144; do
  timeout 900 python bench.py --blocks $B --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench_$B.err | tee gpurun_out/bench_$B.json
done
