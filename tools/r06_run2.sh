#!/bin/bash
# Round-6 session 2 on the GPU box: the GPU suite, the Winograd gate-B probe, the decoder-beside-convolutions stamps.
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r06
mkdir -p "$out"
cd "$root"
timeout 600 tools/build/wino_probe > "$out/wino_probe.log" 2>&1
cat "$out/wino_probe.log"
OG_DECODER_LIB=$root/tools/build/libog_stamps.so timeout 600 python tools/decoder_contention.py > "$out/decoder_contention.log" 2>&1
tail -40 "$out/decoder_contention.log"
timeout 1500 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1
tail -5 "$out/pytest_gpu.log"
