# After `gpurun -- bash scripts/gpu_r6_final.sh`: gpurun_out/final + gpurun_out/traffic -> the tracked files under profiles/ (round 6)
set -e
cd "$(dirname "$0")/.."
python scripts/traffic_merge.py r06 > /tmp/traffic_merge.log; tail -3 /tmp/traffic_merge.log
for c in cfg1 cfg1_strict cfg2_bf16 cfg3_per_gpu_f32 cfg4_edit; do cp gpurun_out/final/trace_$c.txt profiles/trace_latest_$c.txt; cp gpurun_out/final/trace_$c.txt profiles/r06_kernel_trace_$c.txt; done
{ echo "# python -m pytest tests -m gpu -q --durations=15 -s on MI355X (scripts/gpu_r6_final.sh), sources $(python -c 'import bench; print(bench.source_hash())'), git $(git rev-parse --short HEAD)"; grep -vE "^\s*$" gpurun_out/final/suite.log | grep -E "passed|failed|skipped|slowest|s (call|setup)|exit=" ; tail -1 gpurun_out/final/smoke.log; } > profiles/r06_gpu_suite.log
{ echo "# scripts/debug_clocks.py 2 600 on MI355X, library built with -DSAID_CLK_STAMPS (round 6 sources): shader-clock stamps of workgroup 8 (last sample, slice 0) of every launch of one UNet evaluation (B = 2, T = 600, forward schedule)"; cat gpurun_out/final/phase_clocks_b1.txt; } > profiles/r06_phase_clocks_b1.txt
grep '^{' gpurun_out/final/bench_default.log | tail -1 > profiles/r06_bench_default_with_secondary.json
grep '^{' gpurun_out/final/bench_driver_like.log | tail -1 > profiles/r06_bench_driver_like_steps20_warmup5.json
python scripts/index_profiles.py > /dev/null 2>&1 || true
python - <<'PY'
import json
d = json.load(open("profiles/r06_bench_default_with_secondary.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_stale"))
for k, v in d.get("secondary", {}).items():
    print(k, v.get("value"), v.get("ms_per_denoise_step", v.get("ms_per_step")), v.get("roofline", {}).get("kernel"), v.get("roofline", {}).get("unet_step", {}).get("hbm_frac"))
PY
