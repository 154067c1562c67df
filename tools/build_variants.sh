#!/bin/bash
# Kernel A/B builds: tools/build/libog_exp<N>.so = the product library with conv3x3.hip compiled -DHALO_EXP=<N>.
# Use with OG_DECODER_LIB=tools/build/libog_exp<N>.so.
set -e
cd "$(dirname "$0")/.."
python -m offsetguided_amd.build >/dev/null
mkdir -p tools/build
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off --offload-arch=gfx950 -DHALO_EXP=$n \
    -I include -I offsetguided_amd/csrc -x hip -c offsetguided_amd/csrc/conv3x3.hip -o tools/build/conv3x3_exp$n.o
  objs=$(ls offsetguided_amd/build/*.o | grep -v conv3x3.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/build/libog_exp$n.so $objs tools/build/conv3x3_exp$n.o
  echo built tools/build/libog_exp$n.so
done
