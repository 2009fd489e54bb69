#!/bin/bash
# PMC passes over one GEMM shape:  tools/pmc/one_shape.sh <shape-substring> <tune> <outdir> <group>...
# Each group is one rocprofv3 process; prints the last launch's counters of the GEMM kernel.
SHAPE=$1; TUNE=$2; OUT=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
g=0
for grp in "$@"; do
  d="$OUT/g$g"
  timeout -k 5 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$d" -- python3 "$REPO/tools/gemm_bench.py" --only "$SHAPE" --reps 3 --tune "$TUNE" > "$d.log" 2>&1 || { echo "pass failed: g$g"; tail -3 "$d.log"; }
  g=$((g+1))
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for path in sorted(glob.glob(f'{sys.argv[1]}/g*/*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if 'sgemm_glds' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for name, vals in acc.items():
        print(f'{name:36s} {vals[-1]:16.5g}')
PY
