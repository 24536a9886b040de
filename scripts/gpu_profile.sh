# Round profile: kernel trace + stats, then HBM traffic counters in separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
B=${1:-6144}
TAG=${2:-r01}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python3 $R/bench.py --blocks $B --steps 3 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py --blocks $B --steps 1 --warmup 0 --no-cpu-baseline > $O/fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py --blocks $B --steps 1 --warmup 0 --no-cpu-baseline > $O/write.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/sq1 -o s --output-format csv -- python3 $R/bench.py --blocks $B --steps 1 --warmup 0 --no-cpu-baseline > $O/sq1.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/sq2 -o s --output-format csv -- python3 $R/bench.py --blocks $B --steps 1 --warmup 0 --no-cpu-baseline > $O/sq2.log 2>&1
find $O -name "*.csv" | head -30
tail -2 $O/trace.log
