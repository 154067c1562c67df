#!/bin/bash
# Kernel A/B builds: tools/build/libog_<tag>.so = the product library with ONE source file (and its fp16 twin) compiled with extra -D flags.
#   tools/build_variants.sh conv3x3.hip base "" try1 "-DMY_EXPERIMENT=1"
# Use with OG_DECODER_LIB=$PWD/tools/build/libog_<tag>.so (offsetguided_amd/_lib.py) to compare variants in one gpurun
# session (timings from different sessions / boxes differ by several percent).
set -e
cd "$(dirname "$0")/.."
src=$1; shift
python -m offsetguided_amd.build >/dev/null
mkdir -p tools/build
stem=${src%.*}
while [ $# -ge 2 ]; do
  tag=$1; defs=$2; shift 2
  objs=$(ls offsetguided_amd/build/*.o | grep -v "/$stem.o" | grep -v "/${stem}_f16.o")
  mine=""
  for twin in "" "_f16"; do        # the fp16 twin of the source (-DOG_DT_F16=1), where the product has one
    [ -f offsetguided_amd/build/${stem}${twin}.o ] || continue
    dt=""; [ "$twin" = "_f16" ] && dt="-DOG_DT_F16=1"
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off --offload-arch=gfx950 $defs $dt \
      -I include -I offsetguided_amd/csrc -x hip -c offsetguided_amd/csrc/$src -o tools/build/${stem}${twin}_$tag.o &
    mine="$mine tools/build/${stem}${twin}_$tag.o"
  done
  wait
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/build/libog_$tag.so $objs $mine
  echo built tools/build/libog_$tag.so
done
