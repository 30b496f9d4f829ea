"""CPU: the host-side schedule of the matrix-core mel phase of fft512c_kernel (lhotse_amd/csrc/mel4_schedule.hpp) --
emulated instruction by instruction (v_mfma_f32_4x4x1_16B_f32 block layout, row_shr DPP reductions, output columns)
from the very tables the plan uploads, against the dense product with the reference's filterbank
(Wav2LogFilterBank, lhotse/features/kaldi/layers.py:565-578)."""
import ctypes
import os
import subprocess
import tempfile

import numpy as np
import pytest

from lhotse_amd import constants as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RS = 272  # kCPRowStride


@pytest.fixture(scope="module")
def lib():
    out = os.path.join(tempfile.mkdtemp(prefix="mel4_"), "libmel4.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(ROOT, "tests", "native", "mel4_schedule_capi.cpp"), "-o", out])
    return ctypes.CDLL(out)


def build(lib, mel, max_sets=2, max_steps=16):
    K, M = mel.shape
    mel = np.ascontiguousarray(mel, dtype=np.float32)
    nsets = ctypes.c_int(0)
    steps = (ctypes.c_int * 4)()
    step0 = (ctypes.c_int * 4)()
    wtab = np.zeros(64 * 64, dtype=np.float32)
    ltab = np.zeros(4 * 256, dtype=np.float32)
    n = lib.mel4_build(mel.ctypes.data_as(ctypes.c_void_p), M, K, RS, max_sets, max_steps, ctypes.byref(nsets), steps, step0,
                       wtab.ctypes.data_as(ctypes.c_void_p), wtab.size, ltab.ctypes.data_as(ctypes.c_void_p), ltab.size)
    if n <= 0:
        return None
    return nsets.value, list(steps)[: nsets.value], list(step0)[: nsets.value], wtab[:n].reshape(-1, 64, 4), ltab[: nsets.value * 256].reshape(nsets.value, 64, 4)


def emulate(sched, P, M):
    """P: (4 frames, RS) power rows of one wave -> (4, M) mel energies, exactly as the kernel computes them."""
    nsets, steps, step0, wtab, ltab = sched
    out = np.full((4, M), np.nan, dtype=np.float64)
    seen = np.zeros((4, M), dtype=int)
    flat = P.reshape(-1)
    for s in range(nsets):
        poff = ltab[s, :, 0].view(np.int32)
        col = ltab[s, :, 1].view(np.int32)
        m4, m8 = ltab[s, :, 2], ltab[s, :, 3]
        acc = np.zeros((64, 4), dtype=np.float32)  # [lane = 4 slot + filter][register = frame]
        for t in range(steps[s]):
            step = step0[s] + t
            a = flat[poff + t]  # lane = 4 slot + frame
            b = wtab[step // 4, :, step % 4]  # lane = 4 slot + filter
            for blk in range(16):
                acc[4 * blk : 4 * blk + 4, :] += np.outer(b[4 * blk : 4 * blk + 4], a[4 * blk : 4 * blk + 4]).astype(np.float32)
        for i in range(4):
            v = acc[:, i].copy()
            sh = np.zeros(64, dtype=np.float32)
            for l in range(64):
                sh[l] = v[l - 4] if (l & 15) >= 4 else 0.0
            v = v + sh * m4
            for l in range(64):
                sh[l] = v[l - 8] if (l & 15) >= 8 else 0.0
            v = v + sh * m8
            for l in range(64):
                if col[l] < M:
                    out[i, col[l]] = v[l]
                    seen[i, col[l]] += 1
    return out, seen


@pytest.mark.parametrize("M,sr", [(80, 16000), (64, 16000), (72, 16000), (80, 16000.0)])
def test_schedule_reproduces_the_dense_filterbank_product(lib, M, sr):
    mel = np.asarray(C.make_kaldi_mel(M, 512, sr, 20.0, -400.0), dtype=np.float32)  # (257, M)
    sched = build(lib, mel)
    assert sched is not None, "the default filterbanks must fit the static schedule"
    nsets, steps, step0, wtab, ltab = sched
    assert nsets <= 2 and all(s % 4 == 0 and 0 < s <= 16 for s in steps)
    # operand reads stay inside the wave's four power rows and are 16-byte aligned
    poff = ltab[:, :, 0].view(np.int32)
    assert (poff % 4 == 0).all() and (poff >= 0).all()
    for s in range(nsets):
        assert (poff[s] + steps[s] <= 4 * RS).all()
    rs = np.random.RandomState(0)
    P = np.zeros((4, RS), dtype=np.float32)
    P[:, :257] = rs.rand(4, 257).astype(np.float32) ** 4 * 50.0
    got, seen = emulate(sched, P, M)
    assert (seen == 1).all(), "every (frame, filter) must be written exactly once"
    want = P[:, :257].astype(np.float64) @ mel.astype(np.float64)
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


def test_filterbanks_outside_the_static_schedule_are_refused(lib):
    mel = np.asarray(C.make_kaldi_mel(23, 512, 16000, 20.0, -400.0), dtype=np.float32)  # bands of up to 89 bins per group of 4
    assert build(lib, mel) is None


def test_a_operand_reads_spread_over_bank_quads(lib):
    """A 16-lane group of the kernels' 16-byte power-row reads (4 slots x 4 frames, frame rows 16 banks apart) is conflict-free only if
    its four slots start on four different bank quads (mod 16 dwords); tools/ubench/lds_rate.hip: 3.9 / 7.1 / 12.1 clk per instruction for
    1 / 2 / 4 slots per quad.  The placement keeps the step-weighted worst multiplicity per row low (it was 2.3 before it looked)."""
    mel = np.asarray(C.make_kaldi_mel(80, 512, 16000, 20.0, -400.0), dtype=np.float32)
    nsets, steps, step0, wtab, ltab = build(lib, mel)
    cost = ideal = 0
    for s in range(nsets):
        poff = ltab[s, :, 0].view(np.int32)
        for row in range(4):
            quads = [int((poff[16 * row + 4 * b] % 16) // 4) for b in range(4)]  # lane 4 b + 0: frame 0 of slot b
            cost += max(quads.count(q) for q in set(quads)) * steps[s]
            ideal += steps[s]
    assert cost / ideal <= 1.7, cost / ideal
