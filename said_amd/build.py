"""Build libsaid_hip.so (the gfx950 engine) in-tree with hipcc.

``python -m said_amd.build`` or ``said_amd.build.build_library()``.  hipcc
cross-compiles for gfx950 without a GPU present; the resulting shared object
lives next to the sources (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsaid_hip.so")
# gfx950 only.  SAID_OFFLOAD_ARCH may narrow the target ID for experiments (e.g. "gfx950:xnack-"; several, comma-separated,
# give a fat binary from which the runtime picks the one matching the device).
ARCHS = os.environ.get("SAID_OFFLOAD_ARCH", "gfx950").split(",")
SOURCES = ["gemm.hip", "gemm_lds.hip", "attn.hip", "attn2q.hip", "misc.hip", "out_sched.hip", "conv_in.hip", "tgemm.hip", "rgemm.hip", "stchain.hip", "engine.cpp"]
FLAGS = [*[f"--offload-arch={a}" for a in ARCHS], "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
# Round 5 (profiles/r05a_pk_fma_hazard.txt): on gfx950 a packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose LOW half
# reads the HIGH register of an operand pair (an op_sel bit set) can read that operand as 0 in lanes 48-63 while ANOTHER wave of the SIMD issues fp16 / bf16
# MFMAs — the silent, concurrency-only corruption round 4 chased as "split-fp16 non-determinism".  hipcc's SLP vectoriser produces exactly that form whenever it
# broadcasts the odd element of a loaded pair ((a, b) coefficients: x * a + b).  These sources are built without SLP vectorisation (bit-identical results: an
# unpacked fma is the same fma), and EVERY object's ISA is scanned: a crossed packed-fp32 operand anywhere fails the build (check_packed_f32 below).
NO_SLP = {"gemm_lds.hip", "misc.hip", "out_sched.hip", "tgemm.hip", "stchain.hip"}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the said_amd engine can only be built with ROCm's hipcc")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


_PK_F32 = None


def crossed_packed_f32(asm_path: str):
    """["kernel: instruction", ...] for every v_pk_{fma,mul,add}_f32 in a device assembly file whose op_sel has a bit set, i.e. whose LOW half reads the HIGH
    register of an operand pair (op_sel_hi = 0, the high half reading a low register, was never seen failing: scripts/ubench/pk_fma_beside_mfma.hip)."""
    global _PK_F32
    import re
    if _PK_F32 is None:
        _PK_F32 = (re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\b.*\bop_sel:\[([01,]+)\]"), re.compile(r"^([A-Za-z_][\w$.]*):"))
    pk, label = _PK_F32
    out, kernel = [], "?"
    with open(asm_path) as f:
        for ln in f:
            m = label.match(ln)
            if m and not m.group(1).startswith(".L"):
                kernel = m.group(1)
            m = pk.match(ln)
            if m and "1" in m.group(2):
                out.append(f"{kernel}: {ln.strip()}")
    return out


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    # every header any source may include: a change to one of them rebuilds all objects
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    headers.append(os.path.join(os.path.dirname(HERE), "include", "said_hip.h"))
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            # leading scalar kernel parameters arrive in SGPRs (14 is what the hardware has room for): see gemm_lds.hip, attn.hip
            extra = ["-mllvm", "-amdgpu-kernarg-preload-count=14"] if (src in ("gemm_lds.hip", "attn.hip", "conv_in.hip", "stchain.hip", "out_sched.hip") and not os.environ.get("SAID_NO_PRELOAD")) else []
            if src == "attn2q.hip":   # MFMA results in the vector registers although a 256-thread workgroup could have 512 registers per wave (attn2q.hip's header)
                extra += ["-mllvm", "-amdgpu-kernarg-preload-count=14", "-mllvm", "-amdgpu-mfma-vgpr-form"]
            extra += os.environ.get("SAID_EXTRA_DEFS", "").split()   # development: -D switches for A/B builds (scripts/gpu_ab_build.sh)
            if src in NO_SLP and not os.environ.get("SAID_KEEP_SLP"):
                extra += ["-fno-slp-vectorize"]
            extra += ["-save-temps=obj"]   # keeps <name>-hip-amdgcn-amd-amdhsa-<arch>.s next to the object: the ISA the checks below read
            # the register allocator's report: a kernel that uses scratch (spills) fails the build — see check_scratch below
            cmd = [hipcc] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage"] + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", obj]
            jobs.append(cmd)

    def check_scratch(cmd, stderr):
        """Round 3 lost most of its work on the bf16 large-batch path to 56-204 bytes of scratch per lane in kernels whose k loops were
        clean: no kernel of this library may spill.  SAID_ALLOW_SCRATCH=1 turns the check into a report."""
        import re
        bad, name = [], None
        for ln in stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                name = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", ln)
            if m and int(m.group(1)) > 0:
                bad.append(f"{name}: {m.group(1)} bytes of scratch per lane")
        if bad:
            msg = "kernels that spill (" + os.path.basename(cmd[-3]) + "):\n  " + "\n  ".join(bad)
            if os.environ.get("SAID_ALLOW_SCRATCH"):
                print(msg, file=sys.stderr)
            else:
                raise RuntimeError(msg + "\n(set SAID_ALLOW_SCRATCH=1 to build anyway)")

    def check_packed_f32(cmd):
        """No packed-fp32 instruction of the library may read a HIGH register for its LOW half (see NO_SLP above).  SAID_ALLOW_CROSSED_PK=1: report only."""
        base = os.path.splitext(cmd[-1])[0]
        bad = []
        for arch in ARCHS:
            asm = f"{base}-hip-amdgcn-amd-amdhsa-{arch.split(':')[0]}.s"
            if not os.path.exists(asm):
                raise RuntimeError(f"{asm} not written: the ISA check needs hipcc's -save-temps=obj output")
            bad += crossed_packed_f32(asm)
        b = os.path.basename(base)
        for f in os.listdir(LIBDIR):   # the bulky temporaries of -save-temps (preprocessed sources, bitcode, host assembly) of THIS source; its device .s stays
            mine = f.startswith(b + "-hip-") or f.startswith(b + "-host-") or (f.startswith(b + ".") and f.endswith(".hipfb"))
            if mine and not (f.startswith(b + "-hip-") and f.endswith(".s")):
                try:
                    os.remove(os.path.join(LIBDIR, f))
                except OSError:
                    pass
        if bad:
            msg = "packed-fp32 instructions with a crossed low-half operand (" + os.path.basename(cmd[-3]) + "):\n  " + "\n  ".join(bad[:20]) + (f"\n  ... {len(bad)} in all" if len(bad) > 20 else "")
            if os.environ.get("SAID_ALLOW_CROSSED_PK"):
                print(msg, file=sys.stderr)
            else:
                raise RuntimeError(msg + "\n(gfx950: such an operand can read as 0 beside another wave's fp16 / bf16 MFMAs — build.py NO_SLP; SAID_ALLOW_CROSSED_PK=1 to build anyway)")

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stderr[-8000:])
        if "-c" in cmd:
            try:
                check_scratch(cmd, r.stderr)
                check_packed_f32(cmd)
            except RuntimeError:
                if os.path.exists(cmd[-1]):
                    os.remove(cmd[-1])   # a re-run must not find (and link) the object of a build that was refused
                raise
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, *[f"--offload-arch={a}" for a in ARCHS], "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
