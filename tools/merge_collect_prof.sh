root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r05_merge
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in synth bench; do
  extra=""; [ $mode = bench ] && extra="--bench-inputs"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$mode -- python3 $root/tools/k1_bench.py --forms two fused --iters 20 $extra > $out/$mode.log 2>&1
  ks=$(ls $out/$mode/*/*_kernel_stats.csv | head -1)
  echo "== $mode inputs"; grep -E "band_topk|merge_collect" $ks | sed "s/(float const.*)\",/\",/; s/(unsigned long const.*)\",/\",/" | cut -d, -f1-4
  rm -rf $out/$mode
done
