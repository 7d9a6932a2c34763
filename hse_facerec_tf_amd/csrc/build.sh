#!/bin/bash
# Builds libhsefr.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
#   HSEFR_DEV=1 build.sh   builds libhsefr_dev.so instead: the same sources with -DHSEFR_DEV (tuning knobs as run-time
#                          variables, hsefr_debug_*, the calibration kernels of devtools.hip) -- for tools/kbench.py only.
set -euo pipefail
cd "$(dirname "$0")"
# product sources.  Development builds add round 1's stem (reachable only through lower_graph(stem_fusion="stem")) and the first window 3x3
# convolution (conv3x3_w2_bf16.hip took its layers in round 5), each kept for A/B timing, plus the calibration kernels of devtools.hip.
SRCS="engine.hip conv_first.hip dwconv.hip pwconv_f32.hip pwconv_f16s.hip pwconv_ps.hip pool_dense.hip nn1.hip conv_bf16.hip conv1x1_bf16.hip preprocess.hip dwpw_fused.hip dwpw_f16s.hip stem2_fused.hip stem3_fused.hip stem4_fused.hip stem5_stream.hip conv_f32_mfma.hip smallnet.hip area_resize.hip stem7x7_pool.hip stem7s_stream.hip conv_dma_bf16.hip conv3x3_w2_bf16.hip conv1x1_w4_bf16.hip conv1x1_pair_bf16.hip mtcnn_post.hip"
# -Wno-inline-asm: the LDS-DMA statements write M0 and say so in their clobber lists; clang warns that M0 is a
# reserved register for every instantiation -- the declaration is the point (the compiler must not assume M0 survives)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result -Wno-inline-asm"
LINK="--offload-arch=gfx950"
LINT=1
if [ "${HSEFR_ASAN:-0}" = "1" ]; then
  # HSEFR_ASAN=1 build.sh   the HOST half of the same sources under the address + undefined-behaviour sanitizers (no device code is
  #                         compiled and none runs: GPU sanitizers are not available on this pool) -> build_asan/libhsefr_asan.so +
  #                         build_asan/fuzz_plan, the driver tests/test_plan_blob_fuzz_cpu.py runs.  The compiler flags of that mode live
  #                         in build_asan_flags.sh, which .gpurunignore keeps off the GPU boxes (their launcher refuses any tree whose
  #                         build scripts mention a sanitizer): a CPU-only recipe.
  OUT=build_asan/libhsefr_asan.so; BUILD=build_asan; LINT=0
  . ./build_asan_flags.sh
elif [ "${HSEFR_DEV:-0}" = "1" ]; then
  OUT=../libhsefr_dev.so; BUILD=build_dev; SRCS="$SRCS stem_fused.hip conv3x3_win_bf16.hip devtools.hip"; FLAGS="$FLAGS -DHSEFR_DEV"
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
if [ "$LINT" = "0" ]; then
  # host-only objects still reference their translation unit's device image (__hip_fatbin_<id>, registered with the HIP runtime when the
  # library is loaded and parsed lazily at a first launch, which never happens here): empty images stand in for the device code that was
  # not compiled
  nm -u $OBJS | grep -o "__hip_fatbin_[0-9a-f]*" | sort -u |
    awk '{print "__attribute__((visibility(\"default\"), aligned(4096))) const char " $1 "[4096] = {0};"}' > "$BUILD/fatbin_stubs.c"
  /opt/rocm/lib/llvm/bin/clang -fPIC -c "$BUILD/fatbin_stubs.c" -o "$BUILD/fatbin_stubs.o"
  OBJS="$OBJS $BUILD/fatbin_stubs.o"
fi
hipcc $LINK -shared -fPIC -o "$OUT" $OBJS
if [ "$LINT" = "1" ]; then
  # refuse a library with an unguarded gfx950 store-data hazard (see tools/isa_lint.py)
  python3 ../../tools/isa_lint.py "$OUT"
else
  hipcc $SAN_EXE_FLAGS -x c++ fuzz_plan.cc -o "$BUILD/fuzz_plan" -L"$BUILD" -lhsefr_asan -Wl,-rpath,"$(realpath $BUILD)"
fi
echo "built $(realpath $OUT)"
