set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r05_k1f
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_fused.json 2> $out/bench_fused.err
ks=$(ls $out/stats/*/*_kernel_stats.csv | head -1)
grep -E "band_topk|merge_|collect_limbs|greedy|bicubic" $ks | sed "s/(float const.*)\",/\",/; s/(unsigned long const.*)\",/\",/" | cut -c1-200
cp $ks $out/fused_kernel_stats.csv
rm -rf $out/stats
