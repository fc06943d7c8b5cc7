# After `gpurun -- 'bash scripts/gpu_full.sh; ...'`: turn gpurun_out/ into the tracked summaries under profiles/
set -e
cd "$(dirname "$0")/.."
python scripts/pmc_summary.py gpurun_out/pmc/pmc_FETCH_SIZE_results.db gpurun_out/pmc/pmc_WRITE_SIZE_results.db profiles/traffic_latest.json > /tmp/pmc.txt
( echo "# rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes) -- python bench.py --steps 1 --warmup 0 --num_steps 40 --no_cpu_baseline --no_roofline"
  echo "# MI355X, B=1 (UNet batch 2), T=600.  Per-launch averages; HBM MB = (2*FETCH_SIZE + WRITE_SIZE) KB (gfx950 wide-read half-count correction, MI355X_MICROARCH.md)."
  echo "# Before the XCD-aware block order (same command, commit 'block double-buffering ...'): ugemm_kernel<1,8,0> FETCH 7056.7 KB WRITE 985.8 KB -> 15.1 MB per launch against 3.4 MB algorithmic."
  head -16 /tmp/pmc.txt ) > profiles/r01c_pmc_hbm_traffic.txt
python scripts/prof_summary.py gpurun_out/prof/r01_results.db > gpurun_out/prof_summary.txt
( echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --num_steps 200 --no_cpu_baseline --no_roofline"
  echo "# MI355X; ugemm_kernel<NB, KS, epilogue, variant bits (1 = 3-tap, 2 = GroupNorm seg 0, 4 = GroupNorm seg 1, 8 = GroupNorm'ed residual, 16 = multi-segment, 32 = multi-block K slice), bf16 multiplies, multi-tile>; summary of the rocpd sqlite output via scripts/prof_summary.py"
  cat gpurun_out/prof_summary.txt ) > profiles/r01c_kernel_trace_variants.txt
grep '^{' gpurun_out/bench_default.log | tail -1 > profiles/r01_bench_default.json
[ -f gpurun_out/bench_cfg3_bf16.log ] && grep '^{' gpurun_out/bench_cfg3_bf16.log | tail -1 > profiles/r01_bench_cfg3_b32_n50_bf16.json
[ -f gpurun_out/bench_cfg3_f32.log ] && grep '^{' gpurun_out/bench_cfg3_f32.log | tail -1 > profiles/r01_bench_cfg3_b32_n50_f32.json
( echo "# scripts/debug_clocks.py on MI355X, library built with -DSAID_CLK_STAMPS (shader-clock stamps of workgroup 8 of every GEMM launch of one UNet evaluation, Be=2, T=600)"
  echo "# columns: request issue | GroupNorm finalize | LayerNorm stats | band loads | stage->LDS | MFMA | barrier skew | LDS reduce | epilogue"
  grep -E "^launch" gpurun_out/clk.log ) > profiles/r01c_phase_clocks_variants.txt
python -c "
import json
for f in ['profiles/r01_bench_default.json','profiles/r01_bench_cfg3_b32_n50_f32.json','profiles/r01_bench_cfg3_b32_n50_bf16.json']:
    d=json.load(open(f)); r=d['roofline']; print(f, d['value'], d['ms_per_step'], d['dtype'], r['frac'], r.get('traffic'), r['avg_launch_us'])"
