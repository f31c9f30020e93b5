"""In-tree build of libqed_splat.so (hipcc, gfx950 only).  (The oracle is pure Python / PyTorch: nothing of it is compiled.)

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


def build_lib(force: bool = False, verbose: bool = False, variant: str = "", extra_flags=()) -> Path:
    """``variant`` / ``extra_flags``: a development build beside the product library (objects under
    build/<variant>/, library lib/libqed_splat_<variant>.so; select it with QED_SPLAT_LIB=...)."""
    LIB_DIR.mkdir(exist_ok=True)
    obj_dir = PKG / "build" / variant if variant else PKG / "build"
    obj_dir.mkdir(exist_ok=True, parents=True)
    lib_path = LIB_DIR / f"libqed_splat_{variant}.so" if variant else LIB_PATH
    headers = [CSRC / "qed_common.h", PKG.parent / "include" / "qed_splat.h"]
    objs = []
    procs = []
    for src in SOURCES:
        s = CSRC / src
        o = obj_dir / (Path(src).stem + ".o")
        objs.append(o)
        if force or _stale(o, [s, *headers]):
            cmd = [HIPCC, *FLAGS, *extra_flags, "-c", str(s), "-o", str(o)]
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
    if force or procs or _stale(lib_path, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib_path), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib_path


if __name__ == "__main__":
    # python -m qed_splatter_amd.build [--force] [--variant NAME -DFLAG ...]
    argv = sys.argv[1:]
    variant = argv[argv.index("--variant") + 1] if "--variant" in argv else ""
    print(build_lib(force="--force" in argv, verbose=True, variant=variant,
                    extra_flags=[a for a in argv if a.startswith("-D") or a.startswith("-m")]))
