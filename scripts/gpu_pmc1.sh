set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
B=${1:-256}
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD -d $R/gpurun_out/pmc_a -o a --output-format csv -- python3 $R/bench.py --blocks $B --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU -d $R/gpurun_out/pmc_b -o b --output-format csv -- python3 $R/bench.py --blocks $B --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_b.log 2>&1
ls $R/gpurun_out/pmc_a $R/gpurun_out/pmc_b
