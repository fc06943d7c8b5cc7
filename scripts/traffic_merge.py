"""After `gpurun -- bash scripts/gpu_r3_traffic.sh`: per-config dominant-kernel HBM traffic -> profiles/traffic_latest.json.

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE count
kilobytes; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B per lane) coalesced reads, which is what these kernels
issue, so reads are doubled: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, launch-weighted over the kernel family's variants.
Stamps the file with bench.source_hash() (what bench.py compares at run time) and the git HEAD it was taken at.
usage: python scripts/traffic_merge.py <round tag, e.g. r03>
"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

import re

# Fallback families (config -> bench.py's name of the dominant kernel family) when no bench line is given; normally the names come
# from the `roofline.kernel` fields of a bench line measured on the same sources (3rd argument, default
# gpurun_out/final/bench_default.log): the dominant kernel changes when the schedule does.
FAMILY = {"cfg1": "ugemm_kernel<NB1,KS8,store>", "cfg2_bf16": "xgemm_kernel<96,bf16,store>",
          "cfg3_per_gpu_f32": "fgemm_kernel<96,store>", "cfg4_edit": "ugemm_kernel<NB2,KS8,store>"}
EPI = {"store": 0, "qkv": 1, "geglu": 2, "band": 3}
XGEMM_EK = {"store": ("0", "4"), "qkv": ("1",), "geglu": ("2",), "band": ("3",)}   # xgemm's epilogue-kind template argument


def matcher(name):
    """bench.py's family name -> predicate over rocprofv3's demangled kernel names."""
    m = re.match(r"(ugemm|cgemm)_kernel<NB(\d+),KS(\d+),(\w+)>", name)
    if m:
        pre = f"{m.group(1)}_kernel<{m.group(2)}, {m.group(3)}, {EPI[m.group(4)]}"
        return lambda k: pre + "," in k or pre + ">" in k
    m = re.match(r"(fgemm|tgemm)_kernel<(\d+),(\w+)>", name)
    if m:   # both are instantiations of fgemm_kernel<NJ, KH, BF, ...>: fp32 (K halves) / bf16
        nj = int(m.group(2)) // 32
        if m.group(1) == "fgemm":   # fp32 operands: <NJ, 2, false, ...> (fp32 MFMAs) or <NJ, 1, false, 2, true> (split-fp16 products, round 4)
            return lambda k: f"fgemm_kernel<{nj}, 2, false" in k or f"fgemm_kernel<{nj}, 1, false" in k
        pre = f"fgemm_kernel<{nj}, 1, true"
        return lambda k: pre in k
    m = re.match(r"xgemm_kernel<(\d+),(bf16|f32),(\w+)>", name)
    if m:
        pre = f"xgemm_kernel<{int(m.group(1)) // 32}, {'true' if m.group(2) == 'bf16' else 'false'},"
        eks = XGEMM_EK[m.group(3)]
        return lambda k: pre in k and k[k.index("xgemm_kernel<"):].rstrip(">").split(", ")[5:6] and k[k.index("xgemm_kernel<"):].split(", ")[5] in eks
    m = re.match(r"rgemm_kernel<(\w+)>", name)
    if m:   # rgemm_kernel<NTAP, EK, ...>: EK 0 store, 1 q/k/v, 2 GEGLU, 4 band
        ek = {"store": "0", "qkv": "1", "geglu": "2", "band": "4"}[m.group(1)]
        return lambda k: "rgemm_kernel<" in k and k[k.index("rgemm_kernel<") + 13:].split(", ")[1] == ek
    m = re.match(r"attn_kernel<D(\d+),KS(\d+)>", name)
    if m:
        pre = f"attn_kernel<{int(m.group(1)) // 32}, {m.group(2)},"
        return lambda k: pre in k
    return lambda k: name.split("<")[0] in k


tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
_hf = os.path.join(ROOT, "gpurun_out", "traffic", "source_hash.txt")     # written on the GPU box by gpu_r3_traffic.sh
SRC_HASH = open(_hf).read().strip() if os.path.exists(_hf) else bench.source_hash()
bench_log = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "final", "bench_default.log")
if os.path.exists(bench_log):
    for ln in open(bench_log):
        if ln.startswith("{"):
            d = json.loads(ln)
            FAMILY["cfg1"] = d["roofline"]["kernel"]
            for k, v in d.get("secondary", {}).items():
                if k in FAMILY:
                    FAMILY[k] = v["roofline"]["kernel"]
            print("dominant kernels from", bench_log, FAMILY)
sha = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
out = {"format": "per-config dominant-kernel HBM traffic from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; "
                 "source_hash = bench.source_hash() of the sources measured", "configs": {}}
lines = []
for cfg, name in FAMILY.items():
    is_fam = matcher(name)
    match = name
    def load(counter):
        fs = glob.glob(os.path.join(ROOT, "gpurun_out", "traffic", "**", f"{cfg}_{counter}_summary.json"), recursive=True)
        return json.load(open(fs[0])) if fs else []
    f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
    if not f or not w:
        print(cfg, "missing counters")
        continue
    n = sum(r["launches"] for r in f if is_fam(r["kernel"]) and r["counter"] == "FETCH_SIZE")
    fk = sum(r["sum"] for r in f if is_fam(r["kernel"]) and r["counter"] == "FETCH_SIZE")
    wk = sum(r["sum"] for r in w if is_fam(r["kernel"]) and r["counter"] == "WRITE_SIZE")
    if not n:
        print(cfg, "kernel family not found:", match)
        continue
    per = (2 * fk + wk) * 1024 / n
    out["configs"][cfg] = {"kernel": name, "rocprof_match": match, "launches": n, "hbm_bytes_per_launch": round(per),
                           "fetch_kb_per_launch": round(fk / n, 1), "write_kb_per_launch": round(wk / n, 1),
                           "note": "(2*FETCH_SIZE + WRITE_SIZE) * 1024, launch-weighted over the family's variants; gfx950 half-count correction on reads",
                           "source": f"profiles/{tag}_pmc_hbm_traffic.txt", "source_hash": SRC_HASH, "git_sha": sha}
    lines.append(f"{cfg:18s} {name:32s} launches {n:6d}  FETCH {fk / n:10.1f} KB  WRITE {wk / n:10.1f} KB  HBM {(per) / 1e6:8.2f} MB per launch")
    # the top SINGLE kernel (one instantiation) of the in-situ trace taken in the same batch, with its own traffic (bench.py's dominant FAMILY can be
    # a sum over several instantiations while the trace's top row is another kernel)
    tr = os.path.join(ROOT, "gpurun_out", "final", f"trace_{cfg}.txt")
    if os.path.exists(tr):
        for ln in open(tr):
            mm = re.match(r"\s*([\d.]+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(said::\S.*)$", ln)
            if mm:
                kname = mm.group(5).strip()
                key = re.sub(r"^said::", "", kname)
                nf = sum(r["launches"] for r in f if r["counter"] == "FETCH_SIZE" and r["kernel"].endswith(key))
                fk1 = sum(r["sum"] for r in f if r["counter"] == "FETCH_SIZE" and r["kernel"].endswith(key))
                wk1 = sum(r["sum"] for r in w if r["counter"] == "WRITE_SIZE" and r["kernel"].endswith(key))
                out["configs"][cfg]["top_single_kernel_in_trace"] = {"kernel": kname, "share_of_kernel_time_pct": float(mm.group(4)), "avg_us_in_situ": float(mm.group(3)),
                                                                    "hbm_bytes_per_launch": round((2 * fk1 + wk1) * 1024 / nf) if nf else None}
                lines.append(f"    top single kernel of the in-situ trace: {kname} ({mm.group(4)} % of kernel time, {mm.group(3)} us): "
                             + (f"{(2 * fk1 + wk1) * 1024 / nf / 1e6:.2f} MB per launch" if nf else "not in the PMC passes"))
                break
    # the rest of the step, for the record
    fam = {}
    for r in f:
        fam.setdefault(r["kernel"], [0, 0.0, 0.0])
        fam[r["kernel"]][0] = r["launches"]; fam[r["kernel"]][1] = r["sum"]
    for r in w:
        if r["kernel"] in fam: fam[r["kernel"]][2] = r["sum"]
    for k, (nn, a, b) in sorted(fam.items(), key=lambda kv: -(2 * kv[1][1] + kv[1][2]))[:10]:
        lines.append(f"    {nn:6d} x {(2 * a + b) * 1024 / nn / 1e6:9.2f} MB  {k[:90]}")
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_latest.json"), "w"), indent=1)
open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_traffic.txt"), "w").write(
    f"# rocprofv3 --pmc FETCH_SIZE --kernel-trace / --pmc WRITE_SIZE --kernel-trace (separate passes) -- python bench.py --steps 1 --warmup 0 "
    f"--no_cpu_baseline --no_roofline --no_secondary <config flags>   (scripts/gpu_r3_traffic.sh; git {sha}, sources {SRC_HASH})\n"
    "# HBM MB per launch = (2*FETCH_SIZE + WRITE_SIZE) KB (gfx950 wide-read half-count correction, MI355X_MICROARCH.md); top 10 kernels per config\n"
    + "\n".join(lines) + "\n")
print("\n".join(lines))
