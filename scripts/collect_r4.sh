# After `gpurun -- bash scripts/gpu_r4_final.sh`: gpurun_out/final + gpurun_out/traffic -> the tracked summaries under profiles/ (tag r04h)
set -e
cd "$(dirname "$0")/.."
python scripts/traffic_merge.py r04h gpurun_out/final/bench_default.log | grep -E "^cfg"
for n in b1 cfg2_b32_bf16 cfg3_b32_f32 cfg4_edit; do ( echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary <flags of $n>   (round 4, scripts/gpu_r4_final.sh, in situ; summary by scripts/prof_summary.py)"; cat gpurun_out/final/trace_$n.txt ) > profiles/r04h_kernel_trace_$n.txt; done
cp gpurun_out/final/suite.log profiles/r04h_gpu_suite.log
grep '^{' gpurun_out/final/bench_default.log | tail -1 > profiles/r04h_bench_default_with_secondary.json
grep '^{' gpurun_out/final/bench_driver_like.log | tail -1 > profiles/r04h_bench_driver_like_steps20_warmup5.json
( echo "# rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --batch 32 --num_steps 20 --dtype bf16 --steps 1 --warmup 0 ... (round 4, configs[2]; scripts/pmc_generic_summary.py)"; cat gpurun_out/final/sq_cfg2.txt ) > profiles/r04h_pmc_sq_cfg2_bf16.txt
( echo "# scripts/rgemm_clocks.py on MI355X (-DSAID_CLK_STAMPS build), 64 x 600 tokens bf16, workgroup 8 of every rgemm launch of one UNet evaluation — AFTER round 4's prologue change"; echo "# (mfma rows: stamp 0 = weight fragments in registers (+offset from the helpers' start), s12 = first barrier passed (relative to stamp 0: negative = before the weights were complete);"; echo "#  helper rows: s11 = first tile requested, s12 = first barrier passed, s1 = second barrier; then per tile [work, work, barrier])"; grep "^launch" gpurun_out/final/rgemm_clocks.txt | cut -c1-260 ) > profiles/r04h_rgemm_clocks.txt
python - <<'PY'
import json
d=json.load(open('profiles/r04h_bench_default_with_secondary.json'))
r=d['roofline']; print('headline', d['value'], d['ms_per_step'], r['unet_step']['ms_loop_per_step'], r['frac'], r['traffic'], d['cpu_baseline']['value'])
for k,v in d['secondary'].items():
    rr=v['roofline']; print(k, v['value'], v['ms_per_step'], v['ms_per_denoise_step'], rr['kernel'], rr['bound'], rr['frac'], rr['traffic'], rr.get('traffic_stale'), rr['avg_launch_us'], rr['unet_step']['mfma_frac'], rr['unet_step'].get('hbm_frac'), rr['audio_encode']['ms_per_clip'])
dd=json.load(open('profiles/r04h_bench_driver_like_steps20_warmup5.json')); print('driver-like', dd['value'], dd['ms_per_step'], dd.get('cfg2_bf16_value'))
PY
