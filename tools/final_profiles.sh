#!/bin/bash
# End-of-round measurement session on the GPU box (one gpurun call): the default bench line, the rocprofv3 kernel statistics of the
# same command, one forward's timeline, K1's HBM traffic (two --pmc passes, kernel-trace only beside them), C1's counter groups, and
# the in-pipeline A/B of PostProcess(fused_upsample).  usage: tools/final_profiles.sh <tag>   -> gpurun_out/<tag>/
set -u
tag=$1
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
timeout 900 python bench.py > "$out/bench.json" 2> "$out/bench.err"
bash tools/ab_env.sh "fused(default):OG_FUSED_UPSAMPLE=1" "k1a+k1:OG_FUSED_UPSAMPLE=0" "fused(default):OG_FUSED_UPSAMPLE=1" "k1a+k1:OG_FUSED_UPSAMPLE=0" "fused(default):OG_FUSED_UPSAMPLE=1" "k1a+k1:OG_FUSED_UPSAMPLE=0" > "$out/fused_upsample_ab.log" 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$out/bench_profiled.json" 2> "$out/bench_profiled.err"
kt=$(ls "$out"/stats/*/*_kernel_trace.csv | head -1)
ks=$(ls "$out"/stats/*/*_kernel_stats.csv | head -1)
cp "$ks" "$out/bench_kernel_stats_raw.csv"
python3 "$root/tools/postfind_stats.py" "$kt" "$out/bench_kernel_stats_postfind.csv"
# (one forward's timeline: tools/timeline_env.sh, with ONE batch in flight -- two interleave their kernels; below)
rm -rf "$out/stats"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_$c" -- python3 "$root/tools/k1_bench.py" --iters 10 --rotate 3 --forms two > "$out/k1_$c.log" 2>&1
done
python3 "$root/tools/pmc_traffic.py" "$out"/pmc_FETCH_SIZE/*/*_counter_collection.csv "$out"/pmc_WRITE_SIZE/*/*_counter_collection.csv "$out/k1_traffic.json" > "$out/k1_traffic.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do cp "$out"/pmc_$c/*/*_counter_collection.csv "$out/k1_pmc_$c.csv"; rm -rf "$out/pmc_$c"; done
cd "$root"
bash tools/timeline_env.sh $tag > /dev/null 2>&1
cp "$root/gpurun_out/${tag}_timeline.csv" "$out/forward_timeline.csv"; cp "$root/gpurun_out/${tag}_timeline_summary.txt" "$out/forward_timeline_summary.txt"
bash tools/prof_c1.sh "gpurun_out/$tag/c1" > "$out/c1_pmc.log" 2>&1
cp "$out/c1/c1_pmc_summary.csv" "$out/c1_pmc_summary.csv" 2>/dev/null
rm -rf "$out/c1"
ls -la "$out"
