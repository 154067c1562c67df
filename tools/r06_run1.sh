#!/bin/bash
# Round-6 evidence session on the GPU box (one gpurun call): the GPU suite, kernel traces of the flip-test bench and of the harness
# (no MIOpen / CK / Tensile kernel may appear: InferenceEngine is strict), and the counter passes of the PRODUCTION decoder kernel
# K1-fused (one counter group per rocprofv3 --pmc pass, --kernel-trace only beside it, the program directly behind `--`).
# usage: tools/r06_run1.sh [notests]   -> gpurun_out/r06/
set -u
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/r06
mkdir -p "$out"
cd "$root"
if [ "${1:-}" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1
  tail -5 "$out/pytest_gpu.log"
fi
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/flip" -- python3 "$root/bench.py" --flip --steps 10 --warmup 3 --no-cpu-baseline --no-extras > "$out/bench_flip_profiled.json" 2> "$out/bench_flip_profiled.err"
cp "$(ls "$out"/flip/*/*_kernel_stats.csv | head -1)" "$out/flip_kernel_stats.csv"; rm -rf "$out/flip"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/harness" -- python3 "$root/tools/harness_profile.py" > "$out/harness_profiled.log" 2>&1
cp "$(ls "$out"/harness/*/*_kernel_stats.csv | head -1)" "$out/harness_kernel_stats.csv"; rm -rf "$out/harness"
pass() {   # tag, counters...
  tag=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/k1f_$tag" -- python3 "$root/tools/k1_bench.py" --forms fused --bench-inputs --iters 10 > "$out/k1f_$tag.log" 2>&1
}
pass valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass wait SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS
pass mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python3 "$root/tools/k1f_pmc_summary.py" "$out" > "$out/k1f_pmc_summary.log" 2>&1
cat "$out/k1f_pmc_summary.log"
rm -rf "$out"/k1f_valu "$out"/k1f_wait "$out"/k1f_mem "$out"/k1f_fetch "$out"/k1f_write
ls -la "$out"
