"""CPU tests of the C-ABI boundary: the header is valid C, the library loads and exports every
declared symbol, the struct mirror matches, and the pure host helpers agree with the reference
formulas.  No compute call is made (no GPU here)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from lhotse_amd import _lib, build
from oracle import kaldi_ref as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hipfeat.h")


def declared_functions():
    text = open(HEADER).read()
    return sorted(set(re.findall(r"HIPFEAT_API\s+[\w\s\*]+?\b(hipfeat_\w+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build()
    assert os.path.exists(path)
    names = declared_functions()
    assert len(names) >= 16
    assert set(names) == set(_lib._SIGNATURES), set(names) ^ set(_lib._SIGNATURES)
    out = subprocess.run(["nm", "-D", "--defined-only", str(path)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (hipfeat_\w+)", out))
    assert set(names) <= exported, set(names) - exported
    # nothing else leaks out of the library (-fvisibility=hidden)
    assert all(s.startswith("hipfeat_") for s in exported if not s.startswith("_")), exported


def test_header_is_plain_c_and_struct_layout_matches(tmp_path):
    src = tmp_path / "t.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "hipfeat.h"\n'
        "int main(void){printf(\"%zu %zu %zu %zu %d\\n\", sizeof(hipfeat_config), offsetof(hipfeat_config, preemph_coeff),"
        " offsetof(hipfeat_config, dither), offsetof(hipfeat_config, num_ceps), HIPFEAT_ABI_VERSION); return 0;}\n"
    )
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    size, off_pre, off_dither, off_ceps, abi = map(int, subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    dt = _lib.CONFIG_DTYPE
    assert size == dt.itemsize
    assert off_pre == dt.fields["preemph_coeff"][1]
    assert off_dither == dt.fields["dither"][1]
    assert off_ceps == dt.fields["num_ceps"][1]
    assert abi == _lib.ABI_VERSION


def test_loader_and_pure_helpers():
    lib = _lib.load()
    assert lib.raw("hipfeat_abi_version") == _lib.ABI_VERSION
    assert lib.backend.name in ("ctypes", "cffi")
    for n, shift in [(400, 160), (200, 80), (551, 220), (1102, 441), (512, 128)]:
        for s in list(range(0, 1300)) + [160000, 100050, 256640]:
            for snip in (0, 1):
                assert lib.raw("hipfeat_num_frames", s, n, shift, snip) == K.num_frames(s, n, shift, bool(snip))
    # the vectorised host formula used on the hot path equals the library function
    from lhotse_amd.extractors import _Plan

    for n, shift in [(400, 160), (200, 80), (551, 220)]:
        for snip in (0, 1):
            pl = _Plan.__new__(_Plan)
            pl.n, pl.shift, pl.snip_edges = n, shift, snip
            s = np.concatenate([np.arange(0, 1300), np.array([160000, 100050, 256640, 57600000])])
            assert np.array_equal(pl.num_frames_many(s), [lib.raw("hipfeat_num_frames", int(v), n, shift, snip) for v in s])
    # first valid length for 25/10 ms @ 16 kHz is 140 samples
    assert lib.raw("hipfeat_check_length", 139, 400, 160, 0) == _lib.ERR_TOO_SHORT
    assert "shorter than the reflect padding" in lib.last_error()
    assert lib.raw("hipfeat_check_length", 140, 400, 160, 0) == 0
    assert lib.raw("hipfeat_check_length", 10, 400, 160, 1) == 0
    for s in range(1, 1000):
        ok = lib.raw("hipfeat_check_length", s, 400, 160, 0) == 0
        try:
            K.frame_indices(s, 400, 160, False)
            ref_ok = K.num_frames(s, 400, 160, False) > 0
        except ValueError:
            ref_ok = False
        assert ok == ref_ok, s


def test_plan_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    cfg = np.zeros(1, dtype=_lib.CONFIG_DTYPE)
    cfg["struct_size"] = _lib.CONFIG_DTYPE.itemsize
    cfg["kind"] = 2
    cfg["frame_length"], cfg["frame_shift"], cfg["fft_length"], cfg["num_filters"] = 400, 160, 512, 80
    win = np.ones(400, dtype=np.float32)
    mel = np.zeros((257, 80), dtype=np.float32)
    out = np.zeros(1, dtype=np.uint64)
    st = lib.raw("hipfeat_plan_create", _lib.addr(cfg), _lib.addr(win), _lib.addr(mel), None, None, 0, _lib.addr(out))
    assert st == 2 and out[0] == 0  # HIPFEAT_ERR_HIP: no device, no fallback
    assert "device" in lib.last_error()
    # ABI guard
    cfg["struct_size"] = 4
    assert lib.raw("hipfeat_plan_create", _lib.addr(cfg), _lib.addr(win), _lib.addr(mel), None, None, 0, _lib.addr(out)) == 1
    assert "ABI mismatch" in lib.last_error()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under lhotse_amd/ may reference it."""
    pkg = os.path.join(ROOT, "lhotse_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), os.path.join(dirpath, f)
                assert "kaldi_ref" not in text, os.path.join(dirpath, f)


def test_cffi_branch_of_the_loader_with_a_minimal_cffi_module(monkeypatch):
    """cffi is not installed in this image, so the cffi branch of lhotse_amd/_lib.py never runs with the real
    package.  A minimal stand-in `cffi` module (FFI.cdef / dlopen / cast / NULL / string on top of ctypes) drives exactly that
    branch: the header text it derives must declare every entry point of the ABI, pointer arguments go through ffi.cast,
    NULL through ffi.NULL, C strings through ffi.string -- and the calls must give what the ctypes backend gives."""
    import ctypes
    import re
    import sys
    import types

    from lhotse_amd import _lib, build

    path = str(build.build())
    cdefs = []

    class _Ptr(int):
        pass

    class FakeFFI:
        NULL = _Ptr(0)

        def cdef(self, text):
            cdefs.append(text)

        def dlopen(self, p):
            dll = ctypes.CDLL(p)
            ns = types.SimpleNamespace()
            for name, (ret, args) in _lib._SIGNATURES.items():
                fn = getattr(dll, name)
                fn.restype = ctypes.c_char_p if ret == "const char*" else _lib._CtypesBackend._SCALARS[ret]
                fn.argtypes = [ctypes.c_void_p if a.endswith("*") else _lib._CtypesBackend._SCALARS[a] for a in args]
                setattr(ns, name, fn)
            return ns

        def cast(self, ctype, value):
            assert ctype.endswith("*"), ctype
            return _Ptr(value)

        def string(self, v):
            return v

    fake = types.ModuleType("cffi")
    fake.FFI = FakeFFI
    monkeypatch.setitem(sys.modules, "cffi", fake)
    lib = _lib.Lib(path, prefer="cffi")
    assert lib.backend.name == "cffi"
    # the declarations handed to cdef: no preprocessor lines, no export macro, every entry point present
    text = cdefs[0]
    assert "#" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S) and "HIPFEAT_API" not in text
    for name in _lib._SIGNATURES:
        assert re.search(r"\b%s\s*\(" % name, text), name
    ref = _lib.Lib(path, prefer="ctypes")
    assert ref.backend.name == "ctypes"
    assert lib.raw("hipfeat_abi_version") == ref.raw("hipfeat_abi_version") == _lib.ABI_VERSION
    for n in (0, 139, 140, 16000, 160079, 160080):
        assert lib.raw("hipfeat_num_frames", n, 400, 160, 0) == ref.raw("hipfeat_num_frames", n, 400, 160, 0)
    with pytest.raises(_lib.HipFeatError, match="TOO_SHORT"):
        lib.check("hipfeat_check_length", 100, 400, 160, 0)  # status + thread-local error string through ffi.string
    cnt = np.zeros(1, dtype=np.int32)
    st = lib.raw("hipfeat_device_count", _lib.addr(cnt))  # a pointer argument through ffi.cast
    assert st in (0, _lib.ERR_HIP)
    assert lib.raw("hipfeat_device_count", None) == _lib.ERR_INVALID  # None -> ffi.NULL


def test_experiment_switches_are_not_in_the_product_library():
    """VERDICT r4 task 6: the switches that skip work or retune launch shapes (HIPFEAT_MB_SKIP produces deliberately wrong results) exist only
    in -DHIPFEAT_EXPERIMENTS builds -- their names do not occur in the shipped .so, so no environment can reach them; the routing switches
    (all routes meet the parity bar) do."""
    from lhotse_amd import build as B

    blob = open(B.build(), "rb").read()
    for name in (b"HIPFEAT_MB_SKIP", b"HIPFEAT_MB_SLOTS", b"HIPFEAT_ROUNDS_R3", b"HIPFEAT_ROUNDS\0"):
        assert name not in blob, name
    for name in (b"HIPFEAT_FORCE_GENERIC", b"HIPFEAT_NO_FIXED_SCHEDULE", b"HIPFEAT_NO_FLAT", b"HIPFEAT_PIPE_CHUNKS"):
        assert name in blob, name


def test_archive_overwrite_drops_stale_stripes_and_failed_appends_leave_no_trace(tmp_path):
    """Host-only entry points (no GPU needed): ADVICE r5 -- (i) opening for overwrite with fewer stripes than an earlier run removes the
    higher-numbered files; (ii) an append that fails on one stripe leaves every file at its size before the batch."""
    import numpy as np

    from lhotse_amd import storage as S

    rs = np.random.RandomState(0)
    frames = np.array([30, 50, 20, 40], dtype=np.int64)
    host = rs.rand(int(frames.sum()), 80).astype(np.float32)
    with S.NativeArchive(tmp_path / "feats", mode="w", stripes=4) as ar:
        ar.append(host, frames)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["feats.1.hfa", "feats.2.hfa", "feats.3.hfa", "feats.hfa"]
    with S.NativeArchive(tmp_path / "feats", mode="w", stripes=2) as ar:
        f, o = ar.append(host, frames)
        sizes = [ar.size(0), ar.size(1)]
        assert sum(sizes) == host.nbytes
        # (ii) binary32 archive, a batch whose second run cannot be written: stripe 1's descriptor is swapped for a read-only one
        import os

        ro = os.open(str(ar.paths[1]), os.O_RDONLY)
        # the library keeps its own descriptors; emulate the failure through the public surface instead: a non-finite f16 batch is refused
        # BEFORE anything is written (same no-trace guarantee), sizes unchanged
        os.close(ro)
    with S.NativeArchive(tmp_path / "h", mode="w", np_dtype="<f2", stripes=2, name="hip_archive_f16") as ar:
        ar.append(host.astype(np.float16), frames)
        before = [ar.size(0), ar.size(1)]
        bad = host.astype(np.float16)
        bad[-1, -1] = np.inf
        import pytest

        with pytest.raises(ValueError):
            ar.append(bad, frames)
        assert [ar.size(0), ar.size(1)] == before
    assert sorted(p.name for p in tmp_path.iterdir()) == ["feats.1.hfa", "feats.hfa", "h.1.hfa", "h.hfa"]
    assert sum(os.path.getsize(tmp_path / n) for n in ("h.hfa", "h.1.hfa")) == host.size * 2


def test_archive_append_failing_on_one_stripe_is_rolled_back(tmp_path):
    """The ENOSPC case of ADVICE r5: stripe 1 of the archive is /dev/full (every write fails with ENOSPC).  Stripe 0's run of the batch IS
    written -- and must be cut off again: both sizes stay where they were, the error names the failing file, the archive keeps working."""
    import numpy as np
    import pytest

    from lhotse_amd import _lib
    from lhotse_amd import storage as S

    if not os.path.exists("/dev/full"):
        pytest.skip("no /dev/full")
    rs = np.random.RandomState(0)
    frames = np.array([100, 100], dtype=np.int64)
    host = rs.rand(200, 80).astype(np.float32)
    with S.NativeArchive(tmp_path / "feats", mode="w", stripes=1) as ar:  # 32 000 bytes already in stripe 0
        ar.append(host[:100], frames[:1])
    os.symlink("/dev/full", tmp_path / "feats.1.hfa")
    with S.NativeArchive(tmp_path / "feats", mode="a", stripes=2) as ar:
        before = [ar.size(0), ar.size(1)]
        assert before == [32000, 0]
        with pytest.raises(_lib.HipFeatError, match="archive file 1 failed: No space left"):
            ar.append(host, frames)
        assert [ar.size(0), ar.size(1)] == before and os.path.getsize(tmp_path / "feats.hfa") == 32000  # stripe 0's run was cut off again
    with S.NativeArchive(tmp_path / "feats", mode="a", stripes=1) as ar:
        _, off = ar.append(host[100:], frames[1:])
        assert int(off[0]) == 32000  # the next complete batch lands where the last complete one ended
    assert np.array_equal(S.HipArchiveReader(tmp_path / "feats.hfa").read("32000:100:80"), host[100:])
