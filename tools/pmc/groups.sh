#!/bin/bash
# Counter passes (one rocprofv3 process per group) over any benchmark script, summarised per kernel:
#   tools/pmc/groups.sh <outdir> <kernel-name regex> <sq|tcc|"group 1;group 2;..."> -- <script.py> [args]
# Prints, per matching kernel, the counters of its LAST launch and its launch count.  The program after -- is started
# as `python3 <script.py> ...` directly under rocprofv3 (no shell, no env: gpurun's rule for --pmc runs).
OUT=$1; FILTER=$2; SET=$3; shift 3
[ "$1" == "--" ] && shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
SCRIPT=$(cd "$REPO" && realpath "$1"); shift
case "$SET" in
  sq)  SET="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC;GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT" ;;
  tcc) SET="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum;TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RD_UNCACHED_32B_sum;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum;FETCH_SIZE;WRITE_SIZE" ;;
esac
cd /tmp && export TMPDIR=/tmp
g=0
IFS=';' read -ra GROUPS_ <<< "$SET"
for grp in "${GROUPS_[@]}"; do
  d="$OUT/g$g"
  timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$d" -- python3 "$SCRIPT" "$@" > "$d.log" 2>&1 || { echo "pass failed: g$g ($grp)"; tail -3 "$d.log"; }
  g=$((g+1))
done
python3 - "$OUT" "$FILTER" <<'PY'
import csv, glob, re, sys, collections
pat = re.compile(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sorted(glob.glob(f'{sys.argv[1]}/g*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name']
        if pat.search(k):
            short = re.sub(r'\(anonymous namespace\)::|void |\(.*$', '', k)
            acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
for short, d in acc.items():
    print(f'## {short}')
    for name, vals in d.items():
        print(f'{name:34s} {vals[-1]:16.6g}   (launches seen: {len(vals)})')
PY
