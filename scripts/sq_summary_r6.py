"""profiles/r06m_sq_lds_l2_counters.txt from the per-pass tables of scripts/gpu_r6_sq.sh (gpurun_out/sq6/<cfg>_{A,B,C}.txt): derived shares per kernel.
usage: python scripts/sq_summary_r6.py > profiles/r06m_sq_lds_l2_counters.txt"""
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

base = "gpurun_out/sq6"


def load(f):
    rows = {}
    lines = open(os.path.join(base, f)).read().splitlines()
    names = lines[0].split()[1:-1]
    for ln in lines[1:]:
        m = re.match(r"\s*(\d+)\s+(.*?)\s{2}(said::.*)$", ln)
        if m:
            rows[m.group(3).strip()] = (int(m.group(1)), dict(zip(names, [float(x) for x in m.group(2).split()])))
    return rows


out = ["# round 6: SQ / LDS / L2 counters of the step kernels, separate rocprofv3 --pmc passes with --kernel-trace only.  Headline and configs[2] tables: sources 8a16a028011a5d12 (BEFORE the",
       f"# GEGLU / battn / tgemm256d changes these counters led to: r06n); configs[3] and configs[4] tables: the final sources ({bench.source_hash()}).",
       "# (scripts/gpu_r6_sq.sh: python bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_roofline --no_secondary + the configuration's flags).  Per-launch averages, summed over the chip.",
       "# Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs;",
       "# kernels run serialised and stretched under counter collection, so shares are quoted, not times.  mfma% = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8);",
       "# wait / istall / active = shares of SQ_WAVE_CYCLES (parked on s_waitcnt or a barrier / issue stall / issuing); valu/mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA;",
       "# ldsconf% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; L2hit% = TCC_HIT / (TCC_HIT + TCC_MISS); L2req MB = TCC_REQ x 128 B."]
for cfg, title in (("cfg1", "headline: 1 clip x 600 frames, guidance (UNet batch 2), fp32 mode on split-fp16 products"),
                   ("cfg2_bf16", "configs[2]: 32 clips x 600 frames, guidance (UNet batch 64), bf16 mode"),
                   ("cfg3_f32", "configs[3] per GPU: 32 clips x 600 frames, guidance, fp32 mode (three clip groups) — scripts/gpu_r6_sq34.sh"),
                   ("cfg4_edit", "configs[4]: 1 clip x 1800 frames, editing, fp32 mode — scripts/gpu_r6_sq34.sh")):
    if not os.path.exists(os.path.join(base, cfg + "_A.txt")):
        continue
    A, B, C = load(cfg + "_A.txt"), load(cfg + "_B.txt"), load(cfg + "_C.txt")
    out += ["", "== " + title,
            f"{'launches':>8} {'mfma%':>6} {'wait%':>6} {'istall%':>7} {'active%':>7} {'valu/mfma':>9} {'ldsconf%':>8} {'L2hit%':>6} {'L2req MB':>9}  kernel"]
    for k, (n, a) in sorted(A.items(), key=lambda kv: -kv[1][0] * kv[1][1]["GRBM_GUI_ACTIVE"]):
        if k not in B or k not in C:
            continue
        b, c = B[k][1], C[k][1]
        wc = max(a["SQ_WAVE_CYCLES"], 1)
        mf = 100 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * a["GRBM_GUI_ACTIVE"] / 8)
        vm = b["SQ_INSTS_VALU"] / b["SQ_INSTS_MFMA"] if b["SQ_INSTS_MFMA"] else float("nan")
        lc = 100 * b["SQ_LDS_BANK_CONFLICT"] / max(b["SQ_LDS_IDX_ACTIVE"], 1)
        hit = 100 * c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
        out.append(f"{n:8d} {mf:6.1f} {100 * a['SQ_WAIT_ANY'] / wc:6.1f} {100 * a['SQ_WAIT_INST_ANY'] / wc:7.1f} {100 * a['SQ_ACTIVE_INST_ANY'] / wc:7.1f} {vm:9.1f} {lc:8.1f} {hit:6.1f} "
                   f"{c['TCC_REQ_sum'] * 128 / 1e6:9.1f}  {k[:70]}")
out += ["",
        "Reading: no step kernel is matrix-bound.  The one-clip step's kernels issue 8-40 vector instructions per matrix instruction and sit parked (memory / barriers) 38-50 % of their wave",
        "cycles: latency chains, as the shader-clock stamps say (r06_phase_clocks_b1.txt).  The bf16 fused tail (configs[2]'s top kernel) issues 16 vector instructions per MFMA with four waves per",
        "SIMD: its vector pipe is asked for about twice the cycles of its matrix pipe (GEGLU's erf, LayerNorm, bf16 packing, the band's softmax); its 1.6 GB of L2 requests per launch hit 94 %.",
        "LDS bank conflicts are 23-54 % of the LDS-active cycles of the channel-major GEMMs (the 2-byte split-plane staging stores); round 5 measured a conflict-free 8-byte staging variant",
        "as no faster (the staging phase waits for the X tile: profiles/r05c_*), so they are recorded here, not claimed as the bound.",
        "tgemm256d_kernel's 46 % WAS a defect (a swizzle term that assumed contiguous ds_read_b128 lane groups): fixed after this pass, r06n (6): 4 %.",
        "configs[3]: prep_kernel's LDS is active 5 % of its launch (its 53 % conflict share is of almost nothing) and it issues ~850 vector instructions per wave: it waits (57 % parked) —",
        "a dependent chain per workgroup with 1.7 waves per SIMD under three clip groups; fgemm_kernel: MFMA busy 11 % of the fp16 pipe, 13 vector instructions per MFMA, L2 hit 83 %."]
print("\n".join(out))
