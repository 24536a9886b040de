# EXPERIMENT (round 4): the duplex step with 8 KiB tables (-DPLZ4_EXP_TABBITS=11: valid LZ4, not liblz4's bytes -- timing only) at
# 5 / 6 / 7 waves per SIMD (10+10, 12+12, 14+14 parser + decoder waves per CU).   bash scripts/exp_duplex_occ.sh <lib>...
R=/root/repo
O=$R/gpurun_out/exp_duplex_occ
mkdir -p $O
for LIB in "$@"; do
  T=$(basename $LIB .so)
  PLZ4HIP_LIB=$R/$LIB timeout -k 10 300 python3 $R/scripts/bench_lib.py --no-cpu-baseline --steps 3 --warmup 1 > $O/$T.json 2> $O/$T.log || exit 1
  python3 -c "import json;d=json.load(open('$O/$T.json'));print('$T', d['value'], d['ms'], d['serial_step']['ms'])"
done
