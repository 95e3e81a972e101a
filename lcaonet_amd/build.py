"""Build liblcaonet_hip.so (hipcc, gfx950 only) in-tree.  `python -m lcaonet_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys
import time

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(CSRC, "liblcaonet_hip.so")
SOURCES = ["lcao_generic.hip", "lcao_mfma.hip", "lcao_lowp.hip", "lcao_node.hip", "lcao_csr.hip", "lcao_neighbors.hip", "lcao_backward.hip", "lcao_double.hip"]


def _newer(a: str, b: str) -> bool:
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True, extra_flags: list[str] | None = None) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(REPO, "include"), "-I", CSRC,
             "-Wall", "-Wno-unused-function"] + (extra_flags or [])
    headers = [os.path.join(CSRC, "lcao_common.h"), os.path.join(CSRC, "lcao_mfma.h"), os.path.join(REPO, "include", "lcaonet_hip.h")]
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in headers):
            cmd = [hipcc, "-c", s, "-o", o] + flags
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    t0 = time.time()
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if procs or not os.path.exists(LIB):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        if verbose:
            print(f"built {LIB} in {time.time() - t0:.1f}s", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
