"""Builds libpatchioner_hip.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

    python patchioner_amd/build.py [--force]

Objects are rebuilt only when a source or header is newer.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libpatchioner_hip.so")
SOURCES = ["api.cpp", "vit_gemm.hip", "vit_gemm256.hip", "vit_gemm_roll.hip", "vit_attention.hip", "vit_fp32.hip", "vit_misc.hip", "region.hip", "project.hip", "decoder.hip", "viecap.hip", "preprocess.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-kernarg-preload-count: the command processor hands the first 16 kernel-argument dwords to the wave in SGPRs instead of the
# wave fetching them from memory as its first instruction (gfx940+).  Every kernel starts ~0.25 us sooner; the decoder is a chain of
# 660 dependent kernels: 4.21 against 4.36 ms per 16-prefix decode, 7.50 against 7.68 at 128 (round 4, same box, two alternations).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-kernarg-preload-count=16", "-I", INCLUDE, "-I", CSRC]


def _newest_header() -> float:
    t = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def _compile(src: str, force: bool, hdr_time: float) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    spath = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(spath), hdr_time):
        return obj
    cmd = [HIPCC] + FLAGS + ["-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, jobs: int = 4) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdr_time = _newest_header()
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr_time), SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
