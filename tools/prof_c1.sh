#!/bin/bash
# Counter passes for C1 (the 160x160 256->256 layer) on the GPU box: one counter group per rocprofv3 --pmc pass, the program
# directly after `--`, every --pmc pass carries --kernel-trace only (no other trace domain).  usage: tools/prof_c1.sh <out dir under gpurun_out> [extra bench args]
set -u
out=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
run() {   # tag, counters...
  tag=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$root/$out/$tag" -- python3 "$root/tools/c1_rounds.py" --shapes 8x160 --dtype bf16 --rounds 2 --reps 4 $EXTRA > "$root/$out/$tag.log" 2>&1
}
EXTRA="$*"
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES
python3 - "$root/$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'conv3x3' not in k: continue
        name = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        acc[(name, r['Counter_Name'])].append(float(r['Counter_Value']))
with open(out + '/c1_pmc_summary.csv', 'w') as f:
    f.write('kernel,counter,mean_per_launch,launches\n')
    for (k, c), v in sorted(acc.items()):
        f.write(f'"{k}",{c},{sum(v)/len(v):.0f},{len(v)}\n')
print(open(out + '/c1_pmc_summary.csv').read())
PY
