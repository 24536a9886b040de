set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k dev_pipeline 2>&1 | tail -5
for B in 256 1024 2560; do
  timeout 900 python bench.py --blocks $B --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/bench_$B.err | tee gpurun_out/bench_$B.json
  tail -2 gpurun_out/bench_$B.err
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r1b -o r1b --output-format csv -- python3 /root/repo/bench.py --blocks 1024 --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/prof_r1b.log 2>&1
ls -R /root/repo/gpurun_out/prof_r1b | head
