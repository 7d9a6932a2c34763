#!/bin/bash
# Knock-out builds of the epilogue GEMM (csrc/pwconv_ps.hip, HSEFR_PS_KO bits: 1 = activation pieces of tiles with channel origin != 0 move no
# bytes, 2 = weight pieces move no bytes, 4 = no epilogue stores; results WRONG, timing only): the development library with that one object
# replaced -> hse_facerec_tf_amd/libhsefr_ko<bits>.so.  usage: bash tools/build_ps_ko.sh 1 3 4 7   (after HSEFR_DEV=1 csrc/build.sh)
set -euo pipefail
cd "$(dirname "$0")/../hse_facerec_tf_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result -Wno-inline-asm -DHSEFR_DEV"
for ko in "$@"; do
  hipcc $FLAGS -DHSEFR_PS_KO=$ko -c pwconv_ps.hip -o build_dev/pwconv_ps_ko$ko.obj &
done
wait
for ko in "$@"; do
  OBJS=$(ls build_dev/*.o | grep -v "/pwconv_ps.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhsefr_ko$ko.so $OBJS build_dev/pwconv_ps_ko$ko.obj
  echo "built libhsefr_ko$ko.so"
done
