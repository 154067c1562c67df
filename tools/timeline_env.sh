#!/bin/bash
# One forward's timeline under an engine switch setting: tools/timeline_env.sh <tag> <ENV=val ...>  -> gpurun_out/<tag>_timeline.csv + summary
# (bench.py --inflight 1 so that one forward's kernels are not interleaved with the next one's)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/tl_$tag
mkdir -p "$out"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$out/stats" -- python3 "$root/bench.py" --steps 8 --warmup 3 --no-cpu-baseline --no-extras --inflight 1 > "$out/bench.json" 2> "$out/bench.err"
kt=$(ls "$out"/stats/*/*_kernel_trace.csv | head -1)
python3 "$root/tools/forward_timeline.py" "$kt" "$root/gpurun_out/${tag}_timeline.csv" > "$root/gpurun_out/${tag}_timeline_summary.txt" 2>&1
rm -rf "$out"
head -3 "$root/gpurun_out/${tag}_timeline_summary.txt"
