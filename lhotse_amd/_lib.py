"""
Thin FFI layer over libhipfeat.so (include/hipfeat.h).

Loader policy: cffi (ABI mode) when the package is importable, otherwise ctypes -- the
FFI convention the reference itself uses (lhotse/tools/libsox.py:74-117).  Both back-ends
see the same convention: every pointer argument is a plain integer address (or None),
every scalar a Python int/float, so the wrappers above this file contain no FFI types.

There is NO CPU fallback: if the library cannot be built/loaded, or a call fails, a
``HipFeatError`` is raised with the library's own message.
"""
from __future__ import annotations

import ctypes
import os
import threading
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import build as _build

ABI_VERSION = 5

# name -> (return C type, [argument C types]) ; mirrors include/hipfeat.h one to one.
_SIGNATURES: Dict[str, Tuple[str, List[str]]] = {
    "hipfeat_abi_version": ("int32_t", []),
    "hipfeat_last_error": ("const char*", []),
    "hipfeat_device_count": ("int", ["int32_t*"]),
    "hipfeat_num_frames": ("int64_t", ["int64_t", "int32_t", "int32_t", "int32_t"]),
    "hipfeat_check_length": ("int", ["int64_t", "int32_t", "int32_t", "int32_t"]),
    "hipfeat_plan_create": (
        "int",
        ["const hipfeat_config*", "const float*", "const float*", "const float*", "const float*", "int32_t", "hipfeat_plan**"],
    ),
    "hipfeat_plan_destroy": ("int", ["hipfeat_plan*"]),
    "hipfeat_plan_feature_dim": ("int32_t", ["const hipfeat_plan*"]),
    "hipfeat_plan_kernel_name": ("const char*", ["const hipfeat_plan*"]),
    "hipfeat_layout_create": (
        "int",
        ["const hipfeat_plan*", "int64_t", "const int64_t*", "const int64_t*", "const int64_t*", "const int64_t*", "int64_t", "void*", "hipfeat_layout**"],
    ),
    "hipfeat_layout_destroy": ("int", ["hipfeat_layout*"]),
    "hipfeat_layout_total_frames": ("int64_t", ["const hipfeat_layout*"]),
    "hipfeat_layout_num_frames": ("int", ["const hipfeat_layout*", "int64_t*"]),
    "hipfeat_extract_layout": ("int", ["const hipfeat_plan*", "const hipfeat_layout*", "const float*", "float*", "void*"]),
    "hipfeat_extract": (
        "int",
        ["const hipfeat_plan*", "const float*", "const int64_t*", "const int64_t*", "const int64_t*", "int64_t", "float*", "const int64_t*", "int64_t", "void*"],
    ),
    "hipfeat_extract_collated": (
        "int",
        ["const hipfeat_plan*", "const float*", "const int64_t*", "const int64_t*", "const int64_t*", "int64_t", "float*", "int64_t", "float",
         "int64_t*", "void*"],
    ),
    "hipfeat_pcm16_to_float": ("int", ["const int16_t*", "float*", "int64_t", "void*"]),
    "hipfeat_float_to_half": ("int", ["const float*", "uint16_t*", "int64_t", "void*"]),
    "hipfeat_global_mvn": ("int", ["const float*", "float*", "const float*", "const float*", "int64_t", "int64_t", "int", "void*"]),
    "hipfeat_specaug": (
        "int",
        ["const float*", "float*", "int64_t", "int64_t", "int64_t", "const hipfeat_warp_segment*", "int64_t", "const hipfeat_mask*", "int64_t", "void*"],
    ),
    "hipfeat_resampler_create": ("int", ["int32_t", "int32_t", "int32_t", "const float*", "int32_t", "hipfeat_resampler**"]),
    "hipfeat_resampler_destroy": ("int", ["hipfeat_resampler*"]),
    "hipfeat_resampled_length": ("int64_t", ["int64_t", "int32_t", "int32_t"]),
    "hipfeat_resample": (
        "int",
        ["const hipfeat_resampler*", "const float*", "const int64_t*", "const int64_t*", "int64_t", "float*", "const int64_t*", "void*"],
    ),
    "hipfeat_speed_bank_create": ("int", ["const hipfeat_resampler* const*", "int32_t", "hipfeat_speed_bank**"]),
    "hipfeat_speed_bank_destroy": ("int", ["hipfeat_speed_bank*"]),
    "hipfeat_minibatch_plan": (
        "int",
        ["hipfeat_speed_bank*", "const hipfeat_plan*", "int64_t", "const int64_t*", "const int64_t*", "const int32_t*", "const int64_t*", "int64_t", "int32_t",
         "int64_t", "const int64_t*", "int64_t*", "int64_t*", "int64_t*", "int64_t*", "int64_t*"],
    ),
    "hipfeat_minibatch_run": ("int", ["hipfeat_speed_bank*", "int64_t", "float*", "int64_t", "float*", "int64_t", "float", "void*"]),
    "hipfeat_archive_open": ("int", ["const char* const*", "int32_t", "int32_t", "hipfeat_archive**"]),
    "hipfeat_archive_append": ("int", ["hipfeat_archive*", "const void*", "int64_t", "const int64_t*", "int32_t", "int32_t", "int32_t*", "int64_t*"]),
    "hipfeat_archive_size": ("int64_t", ["const hipfeat_archive*", "int32_t"]),
    "hipfeat_archive_close": ("int", ["hipfeat_archive*"]),
    "hipfeat_manifest_lines": (
        "int",
        ["const char*", "const int64_t*", "const char*", "const int64_t*", "int64_t", "const int64_t*", "const int64_t*", "const char*", "const int64_t*",
         "int32_t", "const int32_t*", "const int64_t*", "int32_t", "int32_t", "char*", "int64_t", "int64_t*"],
    ),
    "hipfeat_host_pipeline_create": ("int", ["const hipfeat_plan*", "int32_t", "hipfeat_host_pipeline**"]),
    "hipfeat_host_pipeline_destroy": ("int", ["hipfeat_host_pipeline*"]),
    "hipfeat_host_pipeline_submit": (
        "int",
        ["hipfeat_host_pipeline*", "const void* const*", "const int64_t*", "int64_t", "int32_t", "int32_t", "int32_t", "int64_t*", "void**", "int64_t*", "int64_t*"],
    ),
    "hipfeat_host_pipeline_wait": ("int", ["hipfeat_host_pipeline*", "int64_t"]),
    "hipfeat_host_pipeline_release": ("int", ["hipfeat_host_pipeline*", "int64_t"]),
    "hipfeat_host_pipeline_stats": ("int", ["const hipfeat_host_pipeline*", "int64_t*"]),
    "hipfeat_host_register": ("int", ["int32_t", "void*", "int64_t"]),
    "hipfeat_host_unregister": ("int", ["void*"]),
    "hipfeat_host_pipeline_direct_batches": ("int64_t", ["const hipfeat_host_pipeline*"]),
    "hipfeat_extract_host": (
        "int",
        ["const hipfeat_plan*", "const float*", "int64_t", "const int64_t*", "const int64_t*", "const int64_t*", "int64_t", "float*", "int64_t", "const int64_t*", "int64_t", "void*"],
    ),
}

# numpy mirror of `struct hipfeat_config` (include/hipfeat.h); field order and sizes must match.
CONFIG_DTYPE = np.dtype(
    [
        ("struct_size", "<i4"),
        ("kind", "<i4"),
        ("frame_length", "<i4"),
        ("frame_shift", "<i4"),
        ("fft_length", "<i4"),
        ("num_filters", "<i4"),
        ("num_ceps", "<i4"),
        ("snip_edges", "<i4"),
        ("remove_dc_offset", "<i4"),
        ("use_energy", "<i4"),
        ("raw_energy", "<i4"),
        ("use_fft_mag", "<i4"),
        ("apply_lifter", "<i4"),
        ("preemph_coeff", "<f4"),
        ("energy_floor", "<f4"),
        ("mel_floor", "<f4"),
        ("log_offset", "<f4"),
        ("dither", "<f4"),
        ("batch_hop", "<i4"),
    ],
    align=True,
)

# numpy mirrors of `hipfeat_warp_segment` / `hipfeat_mask`
WARP_SEGMENT_DTYPE = np.dtype([("sequence", "<i4"), ("start", "<i4"), ("num_frames", "<i4"), ("center", "<i4"), ("warped", "<i4")])
MASK_DTYPE = np.dtype([("sequence", "<i4"), ("axis", "<i4"), ("begin", "<i4"), ("end", "<i4")])

STATUS_NAMES = {0: "OK", 1: "INVALID", 2: "HIP", 3: "UNSUPPORTED", 4: "TOO_SHORT"}
ERR_INVALID = 1
ERR_HIP = 2
ERR_TOO_SHORT = 4
ERR_UNSUPPORTED = 3


class HipFeatError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libhipfeat: {STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


def _is_ptr(ctype: str) -> bool:
    return ctype.endswith("*")


class _CtypesBackend:
    name = "ctypes"
    _SCALARS = {"int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float}

    def __init__(self, path: str):
        self.dll = ctypes.CDLL(path)
        self.fns = {}
        for name, (ret, args) in _SIGNATURES.items():
            fn = getattr(self.dll, name)  # AttributeError here == a symbol the header declares is missing
            fn.restype = ctypes.c_char_p if ret == "const char*" else self._SCALARS[ret]
            fn.argtypes = [ctypes.c_void_p if _is_ptr(a) else self._SCALARS[a] for a in args]
            self.fns[name] = fn

    def call(self, name: str, *args):
        return self.fns[name](*args)

    def bound(self, name: str):
        return self.fns[name]

    @staticmethod
    def string(v) -> str:
        return v.decode() if v else ""


class _CffiBackend:
    name = "cffi"

    def __init__(self, path: str):
        import cffi  # noqa: F401  (ImportError -> caller falls back to ctypes)

        self.ffi = cffi.FFI()
        header = (_build.PKG.parent / "include" / "hipfeat.h").read_text()
        decl = []
        for line in header.splitlines():
            s = line.strip()
            if s.startswith("#") or s.startswith('extern "C"') or s == "}":
                continue
            decl.append(line.replace("HIPFEAT_API ", ""))
        self.ffi.cdef("\n".join(decl))
        self.dll = self.ffi.dlopen(path)
        self.fns = {name: getattr(self.dll, name) for name in _SIGNATURES}

    def call(self, name: str, *args):
        sig = _SIGNATURES[name][1]
        conv = []
        for a, t in zip(args, sig):
            if _is_ptr(t):
                conv.append(self.ffi.NULL if a is None else self.ffi.cast(t, int(a)))
            else:
                conv.append(a)
        return self.fns[name](*conv)

    def bound(self, name: str):
        return lambda *args: self.call(name, *args)

    def string(self, v) -> str:
        return self.ffi.string(v).decode() if v != self.ffi.NULL else ""


class Lib:
    """Loaded libhipfeat with status checking."""

    def __init__(self, path: str, prefer: Optional[str] = None):
        self.path = path
        backend = None
        if prefer in (None, "cffi"):
            try:
                backend = _CffiBackend(path)
            except ImportError:
                if prefer == "cffi":
                    raise
        if backend is None:
            backend = _CtypesBackend(path)
        self.backend = backend
        v = self.raw("hipfeat_abi_version")
        if v != ABI_VERSION:
            raise HipFeatError(1, f"ABI version {v} of {path} != {ABI_VERSION} expected by the Python host")

    def raw(self, name: str, *args):
        return self.backend.call(name, *args)

    def fn(self, name: str):
        """The bound entry point itself (hot paths that call it per mini-batch skip the name lookup): takes the same plain integers /
        floats / None as `raw` and returns the status code."""
        return self.backend.bound(name)

    def last_error(self) -> str:
        return self.backend.string(self.raw("hipfeat_last_error"))

    def check(self, name: str, *args) -> None:
        st = self.raw(name, *args)
        if st != 0:
            raise HipFeatError(int(st), self.last_error())

    def string(self, name: str, *args) -> str:
        return self.backend.string(self.raw(name, *args))


_lock = threading.Lock()
_lib: Optional[Lib] = None


def load(prefer: Optional[str] = None) -> Lib:
    """Load (building in-tree first if needed).  Raises if neither is possible: the product
    path has no CPU fallback."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch ships its own libamdhip64.so.7; importing it first makes libhipfeat bind to
        # the SAME HIP runtime instance (same SONAME) so device pointers and streams are shared.
        import torch  # noqa: F401

        path = os.environ.get("HIPFEAT_LIB") or str(_build.LIB_PATH)  # HIPFEAT_LIB: experiment builds
        if path == str(_build.LIB_PATH) and _build.needs_build():
            try:
                _build.build()
            except Exception as e:  # stale or missing library and no way to build it
                if not os.path.exists(path):
                    raise HipFeatError(2, f"libhipfeat.so is not built and cannot be built here: {e}") from e
        _lib = Lib(path, prefer=os.environ.get("HIPFEAT_FFI", prefer))
        return _lib


def addr(a: Optional[np.ndarray]) -> Optional[int]:
    """Address of a C-contiguous numpy array (None passes NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def i64(values) -> np.ndarray:
    return np.ascontiguousarray(values, dtype=np.int64)


# ----------------------------------------------------------------------------------------------------------------------------------
# fork() with a live HIP context (round 6)
# ----------------------------------------------------------------------------------------------------------------------------------
# Measured on MI355X / ROCm 7.2 (tools/loader_pipeline_probe.py, profiles/r06_loader_pipeline_probe.txt): while child processes that were
# fork()ed AFTER this process created its HIP context are alive -- a DataLoader's workers -- every host <-> device round trip of the parent
# (upload, launch, download of one 60-cut batch) completes ~30 ms late: 1.35 k cuts/s through the host pipeline instead of 6.7 k.  Polling
# instead of interrupt waits does not help (the work itself finishes late).  Workers that were forked BEFORE the context existed, or started
# by a fork server / spawn, do not have the effect.  lhotse's own batch driver forks its workers when the loop over the DataLoader starts
# (lhotse/cut/set.py:2302-2304, :2374), i.e. before the first extract_batch -- and the Hip* extractors create their plan lazily at that
# first call, so a fresh process is fine.  A process that has used the GPU before (an earlier run, another extractor, torch.cuda) is not.
_PLANS_CREATED = 0
_FORK_WARNED = False


def note_plan_created() -> None:
    global _PLANS_CREATED
    _PLANS_CREATED += 1


def cpu_quota() -> Optional[float]:
    """CPUs' worth of time this container may use (cgroup v2 ``cpu.max``, v1 ``cfs_quota_us``), or None (unlimited / unknown).  The MI355X
    boxes of round 6 report 256 schedulable CPUs and a quota of 16: thread / worker counts sized by the affinity mask alone oversubscribe
    such a host sixteen-fold and are throttled (profiles/r06_host_limits.txt)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        return None if q == "max" else int(q) / int(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def usable_cpus() -> int:
    """min(CPUs this process may run on, the container's CPU quota), at least 1."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cpu_quota()
    return max(1, n if q is None else min(n, int(q + 0.5)))


def hip_live() -> bool:
    """Has this process touched the GPU (a plan of this package, or torch's own context)?"""
    if _PLANS_CREATED:
        return True
    try:
        import torch

        return bool(torch.cuda.is_initialized())
    except Exception:  # noqa: BLE001
        return False


def _before_fork() -> None:
    global _FORK_WARNED
    if _FORK_WARNED or not hip_live() or os.environ.get("HIPFEAT_NO_FORK_WARNING"):
        return
    _FORK_WARNED = True
    import warnings

    warnings.warn(
        "lhotse_amd: this process is fork()ing while it holds a live HIP context.  While the children live (DataLoader workers), every host <-> device "
        "round trip of THIS process was measured ~30 ms slower on MI355X / ROCm 7.2.  Start the workers before the first use of the GPU (the Hip* "
        "extractors create their plan lazily), or use multiprocessing_context='forkserver' for the DataLoader "
        "(lhotse_amd.compute_and_store_features_batch does so on its own when the GPU is already in use).  HIPFEAT_NO_FORK_WARNING=1 silences this.",
        RuntimeWarning, stacklevel=2)


if hasattr(os, "register_at_fork"):
    os.register_at_fork(before=_before_fork)
