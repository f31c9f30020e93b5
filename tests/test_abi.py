"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly
the symbols include/qed_splat.h declares (no compute calls: there is no GPU here)."""
from __future__ import annotations

import os
import re
import subprocess

HEADER = os.path.join(os.path.dirname(os.path.dirname(__file__)), "include", "qed_splat.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(qed_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported(lib):
    from qed_splatter_amd import _lib
    names = _declared()
    assert len(names) >= 14
    cdll = lib._cdll
    for n in names:
        assert hasattr(cdll, n), f"{n} declared in qed_splat.h but not exported"
    # and the ctypes binding table covers every one of them
    assert sorted(_lib.SIGNATURES) == names


def test_exports_are_plain_c(lib):
    from qed_splatter_amd.build import LIB_PATH
    out = subprocess.run(["nm", "-D", "--defined-only", str(LIB_PATH)], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for n in _declared():
        assert n in exported, f"{n} is not an extern \"C\" text symbol"


def test_host_only_entry_points(lib):
    # the library, the header it was built from and the Python binding agree on the ABI version (a direct C-ABI caller
    # built against another header must refuse to go on: argument lists have changed between versions)
    import re
    from qed_splatter_amd import _lib
    header = open(HEADER).read()
    declared = int(re.search(r"#define\s+QED_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.qed_version() == declared == _lib.ABI_VERSION == 3
    # workspace sizing is pure host arithmetic: monotone, and enough for 256 counters per block
    a, b = lib.qed_sort_workspace_bytes(1), lib.qed_sort_workspace_bytes(10_000_000)
    assert 0 < a < b and b >= 256 * 4 * (10_000_000 // 2048)
    assert lib.qed_sort_workspace_bytes(-1) < 0


def test_argument_validation_needs_no_gpu(lib):
    """Invalid arguments are rejected on the host before any launch, with a readable message."""
    rc = lib.qed_composite_fwd(1, 0, 0, 0, 0, 64, 64, 4, 4, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)   # channels = 5
    assert rc == -1 and b"channels" in lib.qed_last_error()
    rc = lib.qed_sort_pairs(0, 0, 0, 0, 0, 100, 0, 0, 0, 0, 0)                     # end_bit = 0
    assert rc == -1 and b"end_bit" in lib.qed_last_error()
    rc = lib.qed_project_fwd(10, 1, 0, 0, 0, 0, 0, 3, 0, 0, 4, 0, 0, 64, 64, 4, 4, 0.3, 0.01, 1e10, 0.0, 0,
                             0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)               # SH degree 4
    assert rc == -1 and b"SH degree" in lib.qed_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from qed_splatter_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setenv("QED_SPLAT_LIB", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.QedSplatError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the library is missing")
    finally:
        monkeypatch.undo()
        _lib._LIB = None
        _lib.load()
