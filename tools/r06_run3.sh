#!/bin/bash
# Round-6 session 3: the tests added since session 2, smoke(), and the default bench line.
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r06
mkdir -p "$out"
cd "$root"
timeout 1200 python -m pytest tests -m gpu -x -q -k "batched_input_chain or folded_flip_beyond or two_rank_bench or strict or no_torch_convolution or resize_and_fused or run_images" > "$out/pytest_new.log" 2>&1
tail -15 "$out/pytest_new.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1
tail -3 "$out/smoke.log"
timeout 900 python bench.py > "$out/bench.json" 2> "$out/bench.err"
tail -c 6000 "$out/bench.json"
tail -5 "$out/bench.err"
