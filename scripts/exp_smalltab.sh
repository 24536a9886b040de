# EXPERIMENT (round 4): how does k_l1_parse scale with parser waves per CU when the table is 8 KiB instead of 16?  A build with
# -DPLZ4_EXP_TABBITS=11 (2048 slots: valid LZ4, not liblz4's bytes -- timing only) runs the serial step with W waves per workgroup.
#   bash scripts/exp_smalltab.sh <lib> <W...>
set -x
R=/root/repo
LIB=$1; shift
O=$R/gpurun_out/exp_smalltab
mkdir -p $O
for W in "$@"; do
  PLZ4HIP_LIB=$LIB PLZ4HIP_EXP_W=$W PLZ4HIP_VERBOSE=1 timeout -k 10 300 python3 $R/scripts/bench_lib.py --duplex 0 --no-cpu-baseline --steps 3 --warmup 1 > $O/W$W.json 2> $O/W$W.log || exit 1
  grep -h "exp:" $O/W$W.log | sort | uniq -c
  python3 -c "import json;d=json.load(open('$O/W$W.json'));print('W=$W', d['ms'], d['config']['stored_ratio'])"
done
