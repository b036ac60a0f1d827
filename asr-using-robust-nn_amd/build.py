"""Builds liblipasr.so for gfx950 with hipcc (cross-compiles without a GPU).

    python asr-using-robust-nn_amd/build.py [--force]

The shared library lands next to the Python package (lipasr/liblipasr.so) so it travels with the
tree; it is git-ignored, never installed into site-packages.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lipasr", "liblipasr.so")
SOURCES = ["core.hip", "spectral.hip", "dense.hip", "optim.hip", "mfcc.hip", "stft_bdft.hip"]
HEADERS = ["common.h", "mlp.h", "mfcc_tables.h", "stft.h", os.path.join("..", "..", "include", "lipasr.h")]
# per-source flags.  stft_bdft.hip: the SLP vectoriser turns the kernel's scalar fp32 chains into v_pk_* with a v_mov per operand
# pair (111 moves per frame), which also cost 39 spilled registers; measured 166 us with it, 118 us without (round 4).
EXTRA = {"stft_bdft.hip": ["-fno-slp-vectorize"]}
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fgpu-rdc" if False else "-fno-gpu-rdc"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return OUT
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        obj = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        cmd = [_hipcc(), *FLAGS, *EXTRA.get(s, []), "-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print(f"hipcc failed on {s}:\n{out}", file=sys.stderr)
    if failed:
        raise RuntimeError("liblipasr build failed")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
