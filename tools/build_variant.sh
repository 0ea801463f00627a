#!/bin/bash
# Experimental build of ONE source with extra defines, linked with the in-tree objects of the
# others: tools/build_variant.sh NAME SOURCE [-DFOO=1 ...]  ->  variants/libasr_NAME.so
# (run with ASR_LIB_PATH=variants/libasr_NAME.so; variants/ is git-ignored, it travels with gpurun)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p variants
base=$(basename "${src%.*}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip "$@" -c asr_study_amd/csrc/$src -o variants/${base}_$name.o
objs=""
for o in asr_study_amd/csrc/*.o; do
  if [ "$(basename $o)" == "$base.o" ]; then objs="$objs variants/${base}_$name.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libasr_$name.so $objs -lpthread -ldl
echo variants/libasr_$name.so
