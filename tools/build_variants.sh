#!/bin/bash
# Kernel A/B builds: tools/build/libog_<tag>.so = the product library with ONE source file compiled with extra -D flags.
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
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off --offload-arch=gfx950 $defs \
    -I include -I offsetguided_amd/csrc -x hip -c offsetguided_amd/csrc/$src -o tools/build/${stem}_$tag.o
  objs=$(ls offsetguided_amd/build/*.o | grep -v "/$stem.o")
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/build/libog_$tag.so $objs tools/build/${stem}_$tag.o
  echo built tools/build/libog_$tag.so
done
