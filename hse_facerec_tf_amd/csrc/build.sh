#!/bin/bash
# Builds libhsefr.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
#   HSEFR_DEV=1 build.sh   builds libhsefr_dev.so instead: the same sources with -DHSEFR_DEV (tuning knobs as run-time
#                          variables, hsefr_debug_*, the calibration kernels of devtools.hip) -- for tools/kbench.py only.
set -euo pipefail
cd "$(dirname "$0")"
SRCS="engine.hip conv_first.hip dwconv.hip pwconv_f32.hip pwconv_f16s.hip pwconv_ps.hip pool_dense.hip nn1.hip conv_bf16.hip conv1x1_bf16.hip preprocess.hip dwpw_fused.hip dwpw_f16s.hip stem_fused.hip stem2_fused.hip stem3_fused.hip stem4_fused.hip stem5_stream.hip conv_f32_mfma.hip smallnet.hip area_resize.hip stem7x7_pool.hip conv_dma_bf16.hip conv3x3_win_bf16.hip conv3x3_w2_bf16.hip conv1x1_w4_bf16.hip mtcnn_post.hip"
# -Wno-inline-asm: the LDS-DMA statements write M0 and say so in their clobber lists; clang warns that M0 is a
# reserved register for every instantiation -- the declaration is the point (the compiler must not assume M0 survives)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result -Wno-inline-asm"
if [ "${HSEFR_DEV:-0}" = "1" ]; then
  OUT=../libhsefr_dev.so; BUILD=build_dev; SRCS="$SRCS devtools.hip"; FLAGS="$FLAGS -DHSEFR_DEV"
else
  OUT=../libhsefr.so; BUILD=build
fi
OBJS=""
PIDS=""
mkdir -p "$BUILD"
for s in $SRCS; do
  o="$BUILD/${s%.hip}.o"
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ common.h -nt "$o" ] || [ hsefr_dev.h -nt "$o" ] || [ ../../include/hsefr.h -nt "$o" ] || [ build.sh -nt "$o" ]; then
    rm -f "$o"
    hipcc $FLAGS ${HSEFR_EXTRA_FLAGS:-} -c "$s" -o "$o" &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $o"
done
for pid in $PIDS; do wait "$pid"; done     # a failed compile fails the build (set -e)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS
# refuse a library with an unguarded gfx950 store-data hazard (see tools/isa_lint.py)
python3 ../../tools/isa_lint.py "$OUT"
echo "built $(realpath $OUT)"
