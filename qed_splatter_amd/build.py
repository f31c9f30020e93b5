"""In-tree build of libqed_splat.so (hipcc, gfx950 only) and of the oracle's C helpers.

`python -m qed_splatter_amd.build` or `build_lib()`; called by `__graft_entry__.build()`.
hipcc cross-compiles without a GPU.  The .so stays in-tree (git-ignored) so it travels to the GPU
box with the snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "lib"
LIB_PATH = LIB_DIR / "libqed_splat.so"
SOURCES = ["project.hip", "isect.hip", "radix_sort.hip", "composite.hip", "loss.hip", "ssim.hip", "metrics.hip", "densify.hip", "backproject.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
    "-Wall", "-Wno-unused-function",
    *os.environ.get("QED_HIPCC_EXTRA", "").split(),     # development builds only (e.g. -DQED_SSIM_TIMING)
]


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> Path:
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = PKG / "build"
    obj_dir.mkdir(exist_ok=True)
    headers = [CSRC / "qed_common.h", PKG.parent / "include" / "qed_splat.h"]
    objs = []
    procs = []
    for src in SOURCES:
        s = CSRC / src
        o = obj_dir / (Path(src).stem + ".o")
        objs.append(o)
        if force or _stale(o, [s, *headers]):
            cmd = [HIPCC, *FLAGS, "-c", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc failed on {src} ---\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    if force or procs or _stale(LIB_PATH, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
