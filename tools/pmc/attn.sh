#!/bin/bash
# SQ counter passes over the fused attention kernels:  tools/pmc/attn.sh <outdir> [attn_bench args]
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
g=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT"; do
  d="$OUT/g$g"
  timeout -k 5 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$d" -- python3 "$REPO/tools/attn_bench.py" --reps 2 "$@" > "$d.log" 2>&1 || { echo "pass failed: g$g"; tail -3 "$d.log"; }
  g=$((g+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for path in sorted(glob.glob(f'{sys.argv[1]}/g*/*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name']
        if 'mha_' in k:
            short = 'fwd' if 'fwd' in k else ('bwd' if 'bwd' in k else 'delta')
            acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
    for short, d in acc.items():
        for name, vals in d.items():
            print(f'{short:6s} {name:32s} {vals[-1]:16.6g}')
PY
