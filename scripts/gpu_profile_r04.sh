# Round-4 profiles (same passes as round 3) (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>/, summarised into profiles/ by
# scripts/summarize_profile.py).  The program follows `--` directly (no env/bash hop); counters in passes of their own.
#   bash scripts/gpu_profile_r03.sh <tag> quick|full <bench args...>
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
TAG=$1; MODE=$2; shift 2
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/bench.py "$@" > $O/trace.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/sq1 -o s --output-format csv -- python3 $R/bench.py "$@" > $O/sq1.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/sq2 -o s --output-format csv -- python3 $R/bench.py "$@" > $O/sq2.log 2>&1
if [ "$MODE" = full ]; then
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py "$@" > $O/fetch.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py "$@" > $O/write.log 2>&1
fi
tail -1 $O/trace.log | cut -c1-300
