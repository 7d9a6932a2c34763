#!/bin/bash
# Knock-out / variant builds of ONE kernel file: the development library with that object rebuilt under -D<macro>=<value>
# -> hse_facerec_tf_amd/libhsefr_<tag><value>.so.   usage: bash tools/build_ko.sh <file.hip> <MACRO> <tag> <v1> <v2> ...   (after HSEFR_DEV=1 csrc/build.sh)
set -euo pipefail
cd "$(dirname "$0")/../hse_facerec_tf_amd/csrc"
SRC=$1; MACRO=$2; TAG=$3; shift 3
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-result -Wno-inline-asm -DHSEFR_DEV"
for v in "$@"; do hipcc $FLAGS -D$MACRO=$v -c $SRC -o build_dev/${SRC%.hip}_$TAG$v.obj & done
wait
for v in "$@"; do
  OBJS=$(ls build_dev/*.o | grep -v "/${SRC%.hip}.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhsefr_$TAG$v.so $OBJS build_dev/${SRC%.hip}_$TAG$v.obj
  echo "built libhsefr_$TAG$v.so"
done
