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
SOURCES = ["gemm.hip", "gemm_lds.hip", "attn.hip", "misc.hip", "out_sched.hip", "conv_in.hip", "tgemm.hip", "rgemm.hip", "engine.cpp"]
FLAGS = [*[f"--offload-arch={a}" for a in ARCHS], "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]


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
            extra = ["-mllvm", "-amdgpu-kernarg-preload-count=14"] if (src in ("gemm_lds.hip", "attn.hip", "conv_in.hip") and not os.environ.get("SAID_NO_PRELOAD")) else []
            extra += os.environ.get("SAID_EXTRA_DEFS", "").split()   # development: -D switches for A/B builds (scripts/gpu_ab_build.sh)
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

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stderr[-8000:])
        if "-c" in cmd:
            try:
                check_scratch(cmd, r.stderr)
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
