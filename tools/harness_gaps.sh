root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r05_harness
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $root/tools/harness_gaps.py --run > $out/run.log 2>&1
kt=$(ls $out/trace/*/*_kernel_trace.csv | head -1)
python3 $root/tools/harness_gaps.py $kt | tee $out/gaps.log
tail -3 $out/run.log
rm -rf $out/trace
