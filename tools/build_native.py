#!/usr/bin/env python3
"""Build every native component in-tree for gfx950 (MI355X).

  libdtf_kernels.so  hand-written HIP kernels + step runtime (hipcc --offload-arch=gfx950)
  libdtf_ckpt.so     TF tensor-bundle (Saver V2) checkpoint codec (host C++)

hipcc cross-compiles without a GPU, so this runs on the CPU-only build box.  The
.so files land in distributed_tensorflow_amd/_lib/ (git-ignored, but shipped to
the GPU box by gpurun).  Objects are cached under build/ keyed by source mtime.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "csrc"
OUT = ROOT / "distributed_tensorflow_amd" / "_lib"
BUILD = ROOT / "build"
ARCH = "gfx950"


def _hipcc() -> str:
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


def _newer(src_files, target: Path) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(s).stat().st_mtime > t for s in src_files)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(map(str, cmd)), r.stdout, r.stderr))
    return r


def build_kernels(force: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    BUILD.mkdir(exist_ok=True)
    OUT.mkdir(parents=True, exist_ok=True)
    headers = [h for d in ("kernels", "runtime") for pat in ("*.hpp", "*.h") for h in (CSRC / d).glob(pat)]
    srcs = sorted((CSRC / "kernels").glob("*.hip")) + sorted((CSRC / "runtime").glob("*.hip"))
    objs = []
    jobs = []
    for s in srcs:
        o = BUILD / (s.stem + ".o")
        objs.append(o)
        if force or _newer([s] + headers, o):
            jobs.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Werror", "-I", str(CSRC / "kernels"),
                         "-c", str(s), "-o", str(o)])
    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(_run, jobs))
    lib = OUT / "libdtf_kernels.so"
    if force or jobs or _newer(objs, lib):
        _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(lib)] + [str(o) for o in objs])
    if verbose:
        print(f"[build] {lib} ({len(jobs)} objects rebuilt)")
    return lib


def build_ckpt(force: bool = False, verbose: bool = True) -> Path:
    OUT.mkdir(parents=True, exist_ok=True)
    srcs = [s for s in sorted((CSRC / "checkpoint").glob("*.cc")) if s.name != "fuzz_main.cc"]   # the sanitizer harness
    hdrs = list((CSRC / "checkpoint").glob("*.h"))
    lib = OUT / "libdtf_ckpt.so"
    if srcs and (force or _newer(srcs + hdrs, lib)):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-Werror", "-o", str(lib)] + [str(s) for s in srcs])
    if verbose and srcs:
        print(f"[build] {lib}")
    return lib


def build_all(force: bool = False, verbose: bool = True):
    return build_kernels(force, verbose), build_ckpt(force, verbose)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
