set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for P in 1 2 4; do
  timeout 900 python bench.py --pipe $P --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench_p$P.err | tee gpurun_out/bench_p$P.json
  tail -2 gpurun_out/bench_p$P.err
done
timeout 900 python bench.py 2>gpurun_out/bench_default.err | tee gpurun_out/bench_default.json
