"""GPU: HipSpecAugment / HipGlobalMVN (hipfeat_specaug, hipfeat_global_mvn) against goldens produced by the reference's
SpecAugment.forward / GlobalMVN on CPU with the same seeds, and against the numpy oracle on other shapes.
GlobalMVN is bit-exact; the bicubic time warp is within a few float32 ulps (2e-5 on values of magnitude ~10); masked
regions carry the sequence mean (float32 sum order differs: 1e-5)."""
import os

import numpy as np
import pytest
import torch

import lhotse_amd as LA
from lhotse_amd import _lib
from lhotse_amd.signal_transforms import apply_specaug
from oracle import specaug_ref as R
from oracle.make_golden_specaug import CASES, make_input, seed_all

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "specaug.npz")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hip_specaug_reproduces_the_reference_for_the_same_seeds(case):
    name, kw, shape, sup, seed = case
    want = np.load(GOLDEN)[name]
    x = make_input(shape, seed)
    tfm = LA.HipSpecAugment(**kw)
    seed_all(seed)
    xd = torch.from_numpy(x).cuda()
    y = tfm(xd, supervision_segments=None if sup is None else torch.tensor(sup, dtype=torch.int32))
    assert y.is_cuda and y.shape == xd.shape and y.data_ptr() != xd.data_ptr()
    assert torch.equal(xd.cpu(), torch.from_numpy(x))  # the input is not modified (features.clone(), :199)
    got = y.cpu().numpy()
    assert np.abs(got - want).max() <= 2e-5, (name, np.abs(got - want).max())
    for b in range(shape[0]):  # sequences the reference left alone (p-check) come back bit for bit
        if np.array_equal(want[b], x[b]):
            assert np.array_equal(got[b], x[b])


@pytest.mark.parametrize("shape", [(1, 1, 1), (3, 1000, 80), (2, 4097, 3), (5, 77, 128), (1, 3000, 80), (2, 50, 5000)])
def test_random_descriptors_against_the_oracle(shape):
    B, T, F = shape
    rng = np.random.RandomState(B * 1000 + T + F)
    x = (rng.randn(B, T, F) * 3 - 8).astype(np.float32)
    segs = []
    for b in range(B):
        if T >= 8 and rng.rand() < 0.8:
            n = int(rng.randint(4, T + 1))
            st = int(rng.randint(0, T - n + 1))
            segs.append((b, st, n, int(rng.randint(1, n)), int(rng.randint(1, n))))
    masks = []
    for b in range(B):
        for _ in range(rng.randint(0, 6)):
            ax = int(rng.randint(1, 3))
            size = T if ax == 1 else F
            lo = int(rng.randint(0, size))
            masks.append((b, ax, lo, int(rng.randint(lo, size + 1))))
    segs = np.array(segs, dtype=_lib.WARP_SEGMENT_DTYPE)
    masks = np.array(masks, dtype=_lib.MASK_DTYPE)
    want = R.apply(x, [segs], masks)
    got = apply_specaug(torch.from_numpy(x).cuda(), [segs], masks).cpu().numpy()
    assert np.abs(got - want).max() <= 2e-5, np.abs(got - want).max()
    # determinism: the partial sums are reduced in a fixed order
    again = apply_specaug(torch.from_numpy(x).cuda(), [segs], masks).cpu().numpy()
    assert np.array_equal(got, again)


def test_argument_checks_and_empty_batches():
    lib = _lib.load()
    x = torch.zeros(2, 10, 4, device="cuda")
    out = torch.empty_like(x)
    bad = np.array([(0, 0, 10, 10, 3)], dtype=_lib.WARP_SEGMENT_DTYPE)  # center == num_frames
    with pytest.raises(_lib.HipFeatError, match="out of range"):
        lib.check("hipfeat_specaug", x.data_ptr(), out.data_ptr(), 2, 10, 4, _lib.addr(bad), 1, None, 0, 0)
    ov = np.array([(1, 0, 6, 2, 3), (1, 5, 5, 2, 3)], dtype=_lib.WARP_SEGMENT_DTYPE)
    with pytest.raises(_lib.HipFeatError, match="overlap"):
        lib.check("hipfeat_specaug", x.data_ptr(), out.data_ptr(), 2, 10, 4, _lib.addr(ov), 2, None, 0, 0)
    with pytest.raises(_lib.HipFeatError, match="distinct"):
        lib.check("hipfeat_specaug", x.data_ptr(), x.data_ptr(), 2, 10, 4, None, 0, None, 0, 0)
    empty = torch.zeros(0, 10, 4, device="cuda")
    assert LA.HipSpecAugment(p=1.0)(empty).shape == (0, 10, 4)
    with pytest.raises(AssertionError, match="single-channel"):
        LA.HipSpecAugment()(torch.zeros(10, 4, device="cuda"))
    with pytest.raises(TypeError):
        LA.HipSpecAugment()(torch.zeros(1, 300, 4, device="cuda", dtype=torch.float64))


def test_global_mvn_is_bit_exact():
    z = np.load(GOLDEN)
    mvn = LA.HipGlobalMVN(80)
    mvn.load_state_dict({"norm_means": torch.from_numpy(z["mvn_means"]), "norm_stds": torch.from_numpy(z["mvn_stds"])})
    x = torch.from_numpy(z["mvn_in"]).cuda()
    assert np.array_equal(mvn(x).cpu().numpy(), z["mvn_forward"])
    assert np.array_equal(mvn.inverse(x).cpu().numpy(), z["mvn_inverse"])
    mvn = mvn.cuda()  # buffers may live on either device
    assert np.array_equal(mvn(x[0]).cpu().numpy(), z["mvn_forward"][0])  # (T, F) input
    assert np.array_equal(mvn(x, None).cpu().numpy(), z["mvn_forward"])
    with pytest.raises(RuntimeError, match="must match"):
        mvn(torch.zeros(2, 3, 40, device="cuda"))


def test_pipeline_fbank_mvn_specaug_stays_on_the_device():
    rng = np.random.RandomState(3)
    xs = [(rng.rand(n).astype(np.float32) - 0.5) for n in (48000, 64000, 32017)]
    feats, lens = LA.HipFbank().extract_collated(xs, 16000)
    mvn = LA.HipGlobalMVN(80)
    tfm = LA.HipSpecAugment(time_warp_factor=20, p=1.0)
    seed_all(5)
    y = tfm(mvn(feats))
    seed_all(5)
    seg_rounds, masks = tfm.draw(*feats.shape, None)
    want = R.apply(feats.cpu().numpy(), seg_rounds, masks)
    assert y.is_cuda and np.abs(y.cpu().numpy() - want).max() <= 2e-5


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_forward_with_random_supervision_segments_against_the_oracle(seed):
    """The full forward (the host's draws + the kernels) on random batches with random, partly overlapping supervision
    segments: same draws replayed through the numpy oracle."""
    rng = np.random.RandomState(100 + seed)
    B, T, F = int(rng.randint(1, 7)), int(rng.randint(120, 900)), int(rng.choice([23, 40, 80]))
    x = (rng.randn(B, T, F) * 3 - 8).astype(np.float32)
    sup = []
    for b in range(B):
        for _ in range(rng.randint(0, 4)):
            st = int(rng.randint(0, T - 20))
            sup.append([b, st, int(rng.randint(10, T))])  # may run past the end of the sequence: sliced like the reference
    sup = torch.tensor(sup, dtype=torch.int32) if sup else None
    tfm = LA.HipSpecAugment(time_warp_factor=int(rng.choice([5, 20, 80])), num_frame_masks=int(rng.randint(1, 6)), p=0.8)
    seed_all(seed)
    y = tfm(torch.from_numpy(x).cuda(), supervision_segments=sup).cpu().numpy()
    seed_all(seed)
    seg_rounds, masks = tfm.draw(B, T, F, sup)
    want = R.apply(x, seg_rounds, masks)
    assert np.abs(y - want).max() <= 2e-5, (np.abs(y - want).max(), len(seg_rounds))


def test_fast_rng_mode_runs_and_is_reproducible():
    x = torch.randn(8, 500, 80, device="cuda") * 3 - 8
    tfm = LA.HipSpecAugment(time_warp_factor=20, p=1.0, fast_rng=True)
    np.random.seed(9)
    a = tfm(x)
    np.random.seed(9)
    b = tfm(x)
    assert torch.equal(a, b) and a.shape == x.shape and not torch.equal(a, x)
    np.random.seed(9)
    seg_rounds, masks = tfm._draw_fast(8, 500, 80)
    want = R.apply(x.cpu().numpy(), seg_rounds, masks)
    assert np.abs(a.cpu().numpy() - want).max() <= 2e-5
