#!/bin/bash
# tools/build/libog_stamps.so = the product library with the decoder's in-kernel time stamps compiled in
# (-DOG_K1_STAMPS: nms_topk.hip, -DOG_K3_STAMPS: group.hip) for tools/decoder_contention.py.  Diagnostic build only.
set -e
cd "$(dirname "$0")/.."
python -m offsetguided_amd.build >/dev/null
mkdir -p tools/build
F="-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off --offload-arch=gfx950 -I include -I offsetguided_amd/csrc -x hip"
/opt/rocm/bin/hipcc $F -DOG_K1_STAMPS=1 -c offsetguided_amd/csrc/nms_topk.hip -o tools/build/nms_topk_stamps.o &
/opt/rocm/bin/hipcc $F -DOG_K3_STAMPS=1 -c offsetguided_amd/csrc/group.hip -o tools/build/group_stamps.o &
wait
objs=$(ls offsetguided_amd/build/*.o | grep -v "/nms_topk.o" | grep -v "/group.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/build/libog_stamps.so $objs tools/build/nms_topk_stamps.o tools/build/group_stamps.o
echo built tools/build/libog_stamps.so
