# Round-2 profiles (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>/, summarised into profiles/ by
# scripts/summarize_profile.py).  The program follows `--` directly (no env/bash hop); counters in passes of their own.
#   bash scripts/gpu_profile_r02.sh l1|c4|c5|hc
set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo
W=${1:-l1}
prof() {  # tag, then the command
    local O=$R/gpurun_out/prof_$1; shift
    mkdir -p $O
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- "$@" > $O/trace.log 2>&1
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- "$@" > $O/fetch.log 2>&1
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- "$@" > $O/write.log 2>&1
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/sq1 -o s --output-format csv -- "$@" > $O/sq1.log 2>&1
    timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/sq2 -o s --output-format csv -- "$@" > $O/sq2.log 2>&1
    tail -1 $O/trace.log | cut -c1-200
}
case $W in
l1) prof r02 python3 $R/bench.py --blocks 6144 --steps 2 --warmup 1 --no-cpu-baseline ;;
c4) prof r02_config4_level12 python3 $R/bench.py --level 12 --blocks 1024 --steps 1 --warmup 1 --no-cpu-baseline
    prof r02_level3 python3 $R/bench.py --level 3 --blocks 1024 --steps 1 --warmup 1 --no-cpu-baseline ;;
hc) prof r02_level9_lists python3 $R/bench.py --level 9 --blocks 1024 --steps 1 --warmup 0 --no-cpu-baseline
    prof r02_level11_lists python3 $R/bench.py --level 11 --blocks 1024 --steps 1 --warmup 0 --no-cpu-baseline ;;
c5) prof r02_config5 python3 $R/scripts/config5_rate.py 512 ;;
esac
