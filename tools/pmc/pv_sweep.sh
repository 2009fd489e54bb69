#!/bin/bash
# PMC passes over one attention GEMM (tools/gemm_bench.py --only <shape>) with the real epilogue, with half of
# the stores and with no epilogue (timing-only ablations 99=32 / 99=8).  One counter group per process.
# usage: tools/pmc/pv_sweep.sh <shape-substring> <outdir>
set -o pipefail
SHAPE=${1:-pv_NN}
OUT=${2:-$GRAFT_REPO_ROOT/gpurun_out/pmc_pv}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
GROUPS_=(
 "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM"
 "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
 "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"
 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"
 "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
 "TCC_HIT_sum TCC_MISS_sum"
 "TCC_NORMAL_WRITEBACK_sum TCC_SRC_FIFO_FULL_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_avr"
)
for variant in ${VARIANTS:-full half none}; do
  case $variant in full) T="9=3";; half) T="9=3,99=32";; none) T="9=3,99=8";; esac
  g=0
  for grp in "${GROUPS_[@]}"; do
    d="$OUT/${variant}_g$g"
    echo "pass $variant g$g: $grp"
    timeout -k 5 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$d" -- python3 "$REPO/tools/gemm_bench.py" --only "$SHAPE" --reps 3 --tune "$T" > "$d.log" 2>&1 || { echo "pass failed: $variant g$g"; tail -5 "$d.log"; }
    g=$((g+1))
  done
  echo "variant $variant done"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
table = collections.OrderedDict()
for variant in ('full', 'half', 'none'):
    for path in sorted(glob.glob(f'{out}/{variant}_g*/*/*counter_collection.csv')):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if 'sgemm_glds' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for name, vals in acc.items():
            vals = vals[len(vals) // 2:]          # skip warm-up launches
            table.setdefault(name, {})[variant] = sum(vals) / len(vals)
with open(os.path.join(out, 'summary.txt'), 'w') as f:
    f.write(f"{'counter':44s} {'full':>16s} {'half':>16s} {'none':>16s}\n")
    for name, row in table.items():
        f.write(f"{name:44s} " + ' '.join(f"{row.get(v, float('nan')):16.4g}" for v in ('full', 'half', 'none')) + '\n')
print(open(os.path.join(out, 'summary.txt')).read())
PY
