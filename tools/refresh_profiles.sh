#!/bin/bash
# One GPU session that regenerates everything under profiles/ for the current build (run through gpurun; copies the
# results into gpurun_out/profiles_new/, to be moved to profiles/ and committed from the build container).
# Round 3: the split-precision modes are measured by bench.py's own alt_math regions only (no microbenchmarks, no
# per-mode sweeps: VERDICT round 2 item 9); new: the exchange path A/B on one GPU, the masked attention timings,
# the decoder line, the HBM-side probe and the TCC request counters of the weight-gradient GEMM.
# NPM_REFRESH_PART=pmc runs only the PMC traffic passes and the default bench line (after a change of the sources that does not
# move any timing: profiles/pmc_traffic.json and the bench line name the build they were taken on).
# NPM_REFRESH_PART=1 | 2 runs one half (a gpurun call is limited to 20 minutes): 1 = PMC traffic, bench, clocks, configs, GEMM shapes,
# row kernels, the dropout line; 2 = attention, parity, exchange path, kernel traces, counters, timelines.
# (Round 5's one-off logs -- exchange shadow, LayerNorm hint placement, row terms from the dctx GEMM -- are not re-taken: that code is unchanged.)
set -o pipefail
PART=${NPM_REFRESH_PART:-all}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
R=${NPM_ROUND:-r06}
OUT=$REPO/gpurun_out/profiles_new
mkdir -p "$OUT"
cd "$REPO"
if [ "$PART" != 2 ]; then
# FIRST the PMC traffic passes: the bench line below quotes their figure (roofline.traffic), so both come from THIS session and
# THIS build (profiles/pmc_traffic.json carries the session name and the sources' id; bench.py refuses any other build's)
echo "== PMC traffic (two passes)"
( cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-alt-math --no-configs > "$OUT/pmc_$c.json" 2> "$OUT/pmc_$c.err" || exit 1
  done ) || exit 1
python3 profiles/summarize_pmc.py "$OUT"/pmc_FETCH_SIZE/*/*_counter_collection.csv "$OUT"/pmc_WRITE_SIZE/*/*_counter_collection.csv --session "${R}:$(date +%Y-%m-%dT%H:%M):$(hostname)" > "$OUT/pmc_traffic.json" || exit 1
cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json
echo "== bench (default command)"; timeout -k 10 500 python bench.py > "$OUT/${R}_bench.json" 2> "$OUT/bench.err" || exit 1
python3 -c "import json; d=json.load(open('$OUT/${R}_bench.json')); t=json.load(open('$OUT/pmc_traffic.json')); assert d['roofline']['traffic'] == t['gemm_family_bytes_per_launch'], (d['roofline']['traffic'], d['roofline'].get('traffic_source')); print('roofline.traffic =', d['roofline']['traffic'], 'from', t['session'])" || exit 1
[ "$PART" == pmc ] && { echo "== done (PMC traffic and the bench line only)"; exit 0; }
echo "== clock / power during 60 steps"; tools/clock_sampler.sh "$OUT/clocks_f32.log" -- timeout -k 10 200 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --no-alt-math --no-configs > "$OUT/bench_60.json" 2>/dev/null
{ echo "bench.py --steps 60 --warmup 3, sysfs freq1_input / power1_input of the loaded card (tools/clock_sampler.sh)";
  echo "math f32: $(python3 -c "import json;d=json.load(open('$OUT/bench_60.json'));print(round(d['value'],1),'samples/s',round(d['ms_per_step'],2),'ms/step')")  $(python3 tools/clock_summary.py $OUT/clocks_f32.log)"; } > "$OUT/${R}_clock_power.log"
echo "== config bench (f32)"; { timeout -k 10 300 python tools/config_bench.py --kernels --cpu; echo "-- TransformerDecoder at size (not a BASELINE config)"; timeout -k 10 200 python tools/config_bench.py --only DEC --kernels;
  echo "-- C3 with the K loop of forward / grad_x by taps (NPM_TUNE 16=0: round 3) and by 16-channel chunks (16=1: default), alternating"; for k in 0 1 0 1; do echo "NPM_TUNE=16=$k"; NPM_TUNE=16=$k timeout -k 10 200 python tools/config_bench.py --only C3 --kernels; done; } > "$OUT/${R}_config_bench.log" 2>&1
echo "== gemm shapes (f32)"; timeout -k 10 200 python tools/gemm_bench.py --tune 10=0 > "$OUT/${R}_gemm_shapes.log" 2>&1
echo "== row kernels"; timeout -k 10 200 python tools/rowops_bench.py > "$OUT/${R}_rowops.log" 2>&1
echo "== round 6: the encoder with drop_rate 0.1 (C5D: device-drawn masks applied inside the LayerNorm kernels) against the same harness without dropout (C5), alternating"
{ echo "# tools/config_bench.py --only C5D / --only C5 --kernels, alternating in one session (both not BASELINE.json configs; C5 = bench.py's headline workload in this harness)";
  for i in 1 2; do for c in C5 C5D; do timeout -k 10 200 python tools/config_bench.py --only $c --kernels --min-seconds 1.0; done; done; } > "$OUT/${R}_dropout_fused.log" 2>&1
fi   # part 1
if [ "$PART" != 1 ]; then
echo "== fused attention core"; { echo "-- saved scores (the default from head size 64 up), then recomputing; medians of 15 back-to-back launches after 30 untimed";
  timeout -k 10 100 python tools/attn_bench.py --save-scores --warm 30 --reps 15; timeout -k 10 100 python tools/attn_bench.py --warm 30 --reps 15;
  echo "-- the 4-wave 32 x 32 x 2 forward (NPM_TUNE 17=0: round 3's), saved scores then recomputing"; timeout -k 10 100 python tools/attn_bench.py --save-scores --warm 30 --reps 15 --tune 17=0; timeout -k 10 100 python tools/attn_bench.py --warm 30 --reps 15 --tune 17=0;
  echo "-- the same with mha_bwd8_kernel for everything (NPM_TUNE 14=3), and with round 3's backward kernels (14=1)"; timeout -k 10 100 python tools/attn_bench.py --save-scores --warm 30 --reps 15 --tune 14=3; timeout -k 10 100 python tools/attn_bench.py --warm 30 --reps 15 --tune 14=1;
  echo "-- head sizes 64 / 32 / 16 (H x D = 1024), saved scores: default (mha_bwd8_kernel) and round 3's 4-wave backward (14=1)";
  for d in 64 32 16; do timeout -k 10 100 python tools/attn_bench.py --save-scores --d $d --h $((1024 / d)) --warm 30 --reps 15; timeout -k 10 100 python tools/attn_bench.py --save-scores --d $d --h $((1024 / d)) --warm 30 --reps 15 --tune 14=1; done;
  echo "-- masked (causal, then random per (b, h)), scores saved (the default) and recomputed; then causal WITHOUT the tile summary (NPM_ATTN_TILE_SKIP=0)"; timeout -k 10 100 python tools/attn_bench.py --save-scores --mask causal --warm 30 --reps 15; timeout -k 10 100 python tools/attn_bench.py --mask causal --warm 30 --reps 15; timeout -k 10 100 python tools/attn_bench.py --save-scores --mask random --warm 30 --reps 15; NPM_ATTN_TILE_SKIP=0 timeout -k 10 100 python tools/attn_bench.py --save-scores --mask causal --warm 30 --reps 15;
  echo "-- masks of different shapes, saved scores (forward / backward of the 4th repetition)"; timeout -k 10 100 python tools/attn_masks.py;
  echo "-- stamps of the 4-wave kernels, scores recomputed"; timeout -k 10 100 python tools/attn_trace.py; echo "-- stamps, scores saved"; timeout -k 10 100 python tools/attn_trace.py --save-scores; } > "$OUT/${R}_attn_core.log" 2>&1
echo "== phase stamps of the shipped backward (diagnostic instance)"; timeout -k 10 200 python tools/attn_trace.py --bwd16 2>&1 | sed -n "/mha_bwd16_kernel, thread 0/,\$p" > "$OUT/${R}_attn_bwd16_phases_trace.log"   # (profiles/r04_attn_bwd16_phases.log = this + two experiments appended by hand)
echo "== attention: saved against recomputed scores per head size"; timeout -k 10 400 bash tools/attn_modes.sh > "$OUT/${R}_attn_modes.log" 2>&1
echo "== parity report (every math mode bench.py reports: f32, bf16x3, f16x2)"; timeout -k 10 900 python tools/parity_report.py > "$OUT/${R}_parity_relative_error.log" 2>&1
echo "== exchange path on one GPU: bench.py without and with a one-rank RCCL communicator (NPM_FORCE_RCCL=1)"
{ echo "bench.py --steps 30 --warmup 5 --no-alt-math --no-configs --no-cpu-baseline, alternating; exchange = rank 0's HIP-event statistics per step";
  for i in 1 2 3; do for f in 0 1; do
    echo "NPM_FORCE_RCCL=$f: $(NPM_FORCE_RCCL=$f timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-alt-math --no-configs --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); e=d.get('exchange',{}); print(round(d['ms_per_step'],3),'ms/step', {k:(round(v,4) if isinstance(v,float) else v) for k,v in e.items() if k in ('bytes_per_step','flushes_per_step','allreduce_ms','exposed_ms','exposed_frac_of_step')})")"
  done; done; } > "$OUT/${R}_exchange_path_ab.log" 2>&1
echo "== masked device launch (HIP_VISIBLE_DEVICES=0 LOCAL_RANK=3)"; HIP_VISIBLE_DEVICES=0 LOCAL_RANK=3 RANK=0 WORLD_SIZE=1 NPM_FORCE_RCCL=1 timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-alt-math --no-configs --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('LOCAL_RANK=3 under HIP_VISIBLE_DEVICES=0:', d['exchange'])" >> "$OUT/${R}_exchange_path_ab.log" 2>&1
echo "== HBM-side probe"; timeout -k 10 200 python tools/hbm_side.py --seconds 2 > "$OUT/${R}_hbm_side.log" 2>&1
echo "== kernel metadata"; python3 tools/kernel_meta.py | sed 's/(anonymous namespace):://g; s/  */ /g' > "$OUT/${R}_kernel_registers.log" 2>&1
echo "== rocprofv3 kernel trace"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-alt-math --no-configs > "$OUT/${R}_bench_under_rocprof.json" 2> "$OUT/prof.err" || exit 1
cd "$REPO"
python3 profiles/summarize_rocprof.py "$OUT"/prof/*/*_kernel_trace.csv --steps 5 --warmup 2 > "$OUT/${R}_bench_kernel_trace.md"
cp "$OUT"/prof/*/*_kernel_stats.csv "$OUT/${R}_bench_kernel_stats.csv"
echo "== rocprofv3 kernel trace of the other configs (C2, C3, C4, DEC)"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_cfg" -- python3 "$REPO/tools/config_bench.py" --min-seconds 0.3 > "$OUT/cfg_under_rocprof.log" 2> "$OUT/prof_cfg.err" && cp "$OUT"/prof_cfg/*/*_kernel_stats.csv "$OUT/${R}_config_kernel_stats.csv"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_dec" -- python3 "$REPO/tools/config_bench.py" --only DEC --min-seconds 0.3 > "$OUT/dec_under_rocprof.log" 2> "$OUT/prof_dec.err" && cp "$OUT"/prof_dec/*/*_kernel_stats.csv "$OUT/${R}_decoder_kernel_stats.csv"
cd "$REPO"
echo "== TCC request counters of the FFN weight-gradient GEMM (DRAM-destined vs all; L2 hit / miss)"
tools/pmc/one_shape.sh "ffn_dw_TN M=1024" 10=0 "$OUT/pmc_tn" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RD_UNCACHED_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum" > "$OUT/${R}_pmc_tcc_ffn_dw.log" 2>&1
echo "== SQ / TCC counters of the kernels that are in the step and in the configs"
{ echo "# tools/pmc/groups.sh: one rocprofv3 --pmc process per counter group, last launch of each kernel"; echo "### attention, C4 shape, saved scores (the default: mha_fwd8_kernel, mha_rowterms_kernel, mha_bwd16_kernel)"; tools/pmc/groups.sh "$OUT/pmc_a1" 'mha_' sq -- tools/attn_bench.py --reps 2 --save-scores; echo "### the same under NPM_TUNE 14=3 (mha_bwd8_kernel)"; tools/pmc/groups.sh "$OUT/pmc_a2" 'mha_bwd' sq -- tools/attn_bench.py --reps 2 --save-scores --tune 14=3; } > "$OUT/${R}_pmc_attn_sq.log" 2>&1
{ echo "### attention, C4 shape, saved scores: L2 requests that leave the XCD"; tools/pmc/groups.sh "$OUT/pmc_a3" 'mha_' tcc -- tools/attn_bench.py --reps 2 --save-scores; } > "$OUT/${R}_pmc_attn_tcc.log" 2>&1
{ for shp in "ffn1_NN" "ffn_dx_NT M=131072 N=1024" "ffn_dw_TN M=1024"; do echo "### gemm_bench --only '$shp'"; tools/pmc/groups.sh "$OUT/pmc_g" 'sgemm_glds' sq -- tools/gemm_bench.py --only "$shp" --reps 2 --tune 10=0; rm -rf "$OUT/pmc_g"; done; } > "$OUT/${R}_pmc_gemm_sq.log" 2>&1
{ echo "### config_bench --only C3 (conv_fwd_glds_kernel<false> = forward, <true> = grad_x, conv_wgrad_relu_kernel)"; tools/pmc/groups.sh "$OUT/pmc_c1" 'conv_' sq -- tools/config_bench.py --only C3 --min-seconds 0.05; } > "$OUT/${R}_pmc_conv_sq.log" 2>&1
{ echo "### config_bench --only C3, K loop by 16-channel chunks (default)"; tools/pmc/groups.sh "$OUT/pmc_c2" 'conv_' tcc -- tools/config_bench.py --only C3 --min-seconds 0.05; echo "### ... by taps (NPM_TUNE=16=0, round 3)"; NPM_TUNE=16=0 tools/pmc/groups.sh "$OUT/pmc_c3" 'conv_fwd' "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum;TCC_HIT_sum TCC_MISS_sum" -- tools/config_bench.py --only C3 --min-seconds 0.05; } > "$OUT/${R}_pmc_conv_tcc.log" 2>&1
echo "== GEMM timeline"; { for a in "1024" "4096" "131072 4096 1024" "131072 128 576" "c2" "qk"; do echo "### gemm_trace.py $a"; timeout -k 10 120 python tools/gemm_trace.py $a; done; } > "$OUT/${R}_gemm_timeline.log" 2>&1
echo "== the reference's own assertion form"; timeout -k 10 300 python tools/reference_form_report.py > "$OUT/${R}_reference_form.md" 2> "$OUT/refform.err"
fi   # part 2
rm -rf "$OUT"/prof "$OUT"/prof_cfg "$OUT"/prof_dec "$OUT"/pmc_FETCH_SIZE "$OUT"/pmc_WRITE_SIZE "$OUT"/pmc_tn "$OUT"/pmc_attn "$OUT"/pmc_a1 "$OUT"/pmc_a2 "$OUT"/pmc_a3 "$OUT"/pmc_g "$OUT"/pmc_c1 "$OUT"/pmc_c2 "$OUT"/pmc_c3
echo "== done"; ls "$OUT"
