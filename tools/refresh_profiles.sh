#!/bin/bash
# One GPU session that regenerates everything under profiles/ for the current build (run through gpurun; copies the
# results into gpurun_out/profiles_new/, to be moved to profiles/ and committed from the build container).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-/root/repo}
R=${NPM_ROUND:-r02}
OUT=$REPO/gpurun_out/profiles_new
mkdir -p "$OUT"
cd "$REPO"
echo "== bench (default command)"; timeout -k 10 400 python bench.py > "$OUT/${R}_bench.json" 2> "$OUT/bench.err" || exit 1
echo "== clock / power during 60 steps"; tools/clock_sampler.sh "$OUT/clocks_f32.log" -- timeout -k 10 200 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-alt-math > "$OUT/bench_60.json" 2>/dev/null
tools/clock_sampler.sh "$OUT/clocks_bf16x3.log" -- timeout -k 10 200 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-alt-math --math bf16x3 > "$OUT/bench_60_bf16x3.json" 2>/dev/null
tools/clock_sampler.sh "$OUT/clocks_f16x2.log" -- timeout -k 10 200 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-alt-math --math f16x2 > "$OUT/bench_60_f16x2.json" 2>/dev/null
{ echo "bench.py --steps 60 --warmup 3, sysfs freq1_input / power1_input of the loaded card (tools/clock_sampler.sh)";
  for m in f32 bf16x3 f16x2; do f=$OUT/bench_60.json; [ $m != f32 ] && f=$OUT/bench_60_$m.json;
    echo "math $m: $(python3 -c "import json;d=json.load(open('$f'));print(round(d['value'],1),'samples/s',round(d['ms_per_step'],2),'ms/step')")  $(python3 tools/clock_summary.py $OUT/clocks_$m.log)"; done; } > "$OUT/${R}_clock_power.log"
echo "== config bench"; { for m in f32 bf16x3 bf16x3_fast f16x2; do echo "NPM_MATH=$m"; NPM_MATH=$m timeout -k 10 300 python tools/config_bench.py --kernels; done; } > "$OUT/${R}_config_bench.log" 2>&1
echo "== gemm shapes"; { for t in 10=0 10=2 10=1 10=3; do timeout -k 10 200 python tools/gemm_bench.py --tune $t; done; } > "$OUT/${R}_gemm_shapes.log" 2>&1
echo "== row kernels"; timeout -k 10 200 python tools/rowops_bench.py > "$OUT/${R}_rowops.log" 2>&1
echo "== math error"; timeout -k 10 100 python tools/math_bias.py > "$OUT/${R}_math_error.log" 2>&1
echo "== fused attention core"; { timeout -k 10 100 python tools/attn_bench.py; timeout -k 10 100 python tools/attn_bench.py --save-scores; echo "-- stamps, scores recomputed"; timeout -k 10 100 python tools/attn_trace.py; echo "-- stamps, scores saved (the default)"; timeout -k 10 100 python tools/attn_trace.py --save-scores; } > "$OUT/${R}_attn_core.log" 2>&1
echo "== GEMM block timelines"; { for sh in "131072 1024 1024" "131072 4096 1024" "131072 1024 4096" "3211264 128 576" c2; do timeout -k 10 100 python tools/gemm_trace.py $sh 2>&1 | grep -E "^K=|^in us|^traced|^matrix pipe"; done; } > "$OUT/${R}_gemm_timeline.log" 2>&1
echo "== scaled fp16 split prototype"; ( cd tools/microbench && { [ -x f16x2_gemm ] || hipcc -O3 --offload-arch=gfx950 f16x2_gemm.hip -o f16x2_gemm; } && timeout -k 10 300 ./f16x2_gemm ) > "$OUT/${R}_f16x2_gemm.log" 2>&1
echo "== split-bf16 prototype and ablations"; ( cd tools/microbench && { [ -x coop_split_gemm ] || hipcc -O3 --offload-arch=gfx950 coop_split_gemm.hip -o coop_split_gemm; } && timeout -k 10 200 ./coop_split_gemm ) > "$OUT/${R}_coop_split_gemm.log" 2>&1
echo "== f32 MFMA issue microbenchmark"; timeout -k 10 60 tools/microbench/mfma_f32_chain > "$OUT/${R}_mfma_f32_chain.log" 2>&1
echo "== parity report"; timeout -k 10 600 python tools/parity_report.py > "$OUT/${R}_parity_relative_error.log" 2>&1
echo "== attention: fused core against the GEMM composition, whole step"; { for cfg in "NPM_ATTN_CORE=0" "NPM_ATTN_CORE=1 NPM_ATTN_SAVE_SCORES=0" "NPM_ATTN_CORE=1 NPM_ATTN_SAVE_SCORES=1"; do echo "$cfg: $(env $cfg timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-alt-math 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1),'samples/s',round(d['ms_per_step'],2),'ms/step')")"; done; } > "$OUT/${R}_attn_step_ab.log" 2>&1
echo "== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-alt-math > "$OUT/${R}_bench_under_rocprof.json" 2> "$OUT/prof.err" || exit 1
cd "$REPO"
python3 profiles/summarize_rocprof.py "$OUT"/prof/*/*_kernel_trace.csv --steps 5 --warmup 2 > "$OUT/${R}_bench_kernel_trace.md"
cp "$OUT"/prof/*/*_kernel_stats.csv "$OUT/${R}_bench_kernel_stats.csv"
echo "== PMC traffic (two passes)"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-alt-math > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err" || exit 1
done
cd "$REPO"
python3 profiles/summarize_pmc.py "$OUT"/pmc_FETCH_SIZE/*/*_counter_collection.csv "$OUT"/pmc_WRITE_SIZE/*/*_counter_collection.csv > "$OUT/pmc_traffic.json"
echo "== SQ counters: split-bf16 FFN GEMM, fused attention"
tools/pmc/one_shape.sh ffn1_NN 10=2 "$OUT/pmc_bf16x3" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES" > "$OUT/${R}_pmc_bf16x3_ffn1.log" 2>&1
tools/pmc/attn.sh "$OUT/pmc_attn" > "$OUT/${R}_pmc_attn_core.log" 2>&1
rm -rf "$OUT"/prof "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_bf16x3 "$OUT"/pmc_attn
echo "== done"; ls "$OUT"
