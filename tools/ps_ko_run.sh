#!/bin/bash
# On the GPU box: the MobileNet-192 and age/gender per-layer tables through the development library and its knock-out builds (tools/build_ps_ko.sh),
# three alternating rounds, one process per run -> gpurun_out/<dir>/ps_ko.txt
OUT=${1:-gpurun_out/ps_ko}
mkdir -p $OUT
: > $OUT/ps_ko.txt
for rnd in 1 2 3; do
  for lib in libhsefr_dev.so libhsefr_ko1.so libhsefr_ko2.so libhsefr_ko3.so libhsefr_ko4.so libhsefr_ko12.so libhsefr_ko7.so; do
    for cfg in mobilenet192 agegender; do
      echo "== round $rnd $lib $cfg" >> $OUT/ps_ko.txt
      HSEFR_LIB=$lib BC_STEPS=30 python tools/bench_configs.py $cfg 2>/dev/null | grep -E "batch|kind 21|kind 22" >> $OUT/ps_ko.txt
    done
  done
done
