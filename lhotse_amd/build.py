"""
In-tree build of libhipfeat.so (HIP kernels + the C ABI of include/hipfeat.h) for gfx950.

    python -m lhotse_amd.build [--force] [-v]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the
working tree.  No torch, no pybind: the library is a plain C-ABI shared object.
"""
from __future__ import annotations

import fcntl
import os
import shutil
import subprocess
import sys
from pathlib import Path
from typing import List

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB_DIR = PKG / "_lib"
LIB_PATH = LIB_DIR / "libhipfeat.so"
SOURCES = [CSRC / "hipfeat.hip"]
ARCH = "gfx950"


def _deps() -> List[Path]:
    return sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.hpp")) + [PKG.parent / "include" / "hipfeat.h"]


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    return any(p.exists() and p.stat().st_mtime > t for p in _deps())


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm); libhipfeat cannot be built")


def build(force: bool = False, verbose: bool = False, extra_flags: List[str] = (), output: Path = None) -> Path:
    """Compile libhipfeat.so.  ``extra_flags`` / ``output`` build experiment variants next to
    the product library (loaded through the HIPFEAT_LIB environment variable)."""
    out_path = Path(output) if output is not None else LIB_PATH
    if output is None and not force and not needs_build():
        return LIB_PATH
    LIB_DIR.mkdir(exist_ok=True)
    # one builder at a time: several ranks of a multi-GPU job may import the package simultaneously
    with open(LIB_DIR / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if output is None and not force and not needs_build():
                return LIB_PATH  # another process built it while we waited
            tmp = f"{out_path}.{os.getpid()}.tmp"
            cmd = [
                hipcc_path(),
                f"--offload-arch={ARCH}",
                "-O3",
                "-std=c++17",
                "-fPIC",
                "-shared",
                "-fvisibility=hidden",
                "-Wall",
                "-Wno-unused-function",
                "-DHIPFEAT_BUILD",
                *list(extra_flags),
                *[str(s) for s in SOURCES],
                "-o",
                tmp,
            ]
            if verbose:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                print(" ".join(cmd), file=sys.stderr)
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError(f"hipcc failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
            if verbose:
                print(res.stderr, file=sys.stderr)
            os.replace(tmp, out_path)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out_path


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose="-v" in sys.argv or "--verbose" in sys.argv)
    print(p)
