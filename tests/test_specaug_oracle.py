"""CPU: the product's random-draw logic (HipSpecAugment.draw: the reference's RNG calls in the reference's order) +
the numpy restatement of the warp / mask arithmetic reproduce the reference's SpecAugment.forward for the same seeds;
GlobalMVN restatement against the reference's output.  Goldens: oracle/make_golden_specaug.py."""
import os

import numpy as np
import pytest
import torch

import lhotse_amd as LA
from oracle import specaug_ref as R
from oracle.make_golden_specaug import CASES, make_input, seed_all

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "specaug.npz")


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_draws_plus_oracle_equal_the_reference(case):
    name, kw, shape, sup, seed = case
    want = np.load(GOLDEN)[name]
    x = make_input(shape, seed)
    tfm = LA.HipSpecAugment(**kw)
    seed_all(seed)
    seg_rounds, masks = tfm.draw(*shape, None if sup is None else torch.tensor(sup, dtype=torch.int32))
    got = R.apply(x, seg_rounds, masks)
    assert got.shape == want.shape
    changed = want != x
    assert np.abs(got - want).max() <= 2e-5, (name, np.abs(got - want).max())  # values are O(10): a few ulps
    if name == "specaug_supervisions":
        assert len(seg_rounds) >= 2  # sequence 3 has overlapping supervisions: applied one after another
    if kw.get("time_warp_factor", 80) is None:
        assert len(seg_rounds) == 0 and np.array_equal(got[~changed], x[~changed])


def test_rng_consumption_matches_the_reference_exactly():
    # after a draw the three generators must be where the reference leaves them: the next numbers are part of the goldens'
    # contract (a training run interleaves many transforms on the same generators)
    import random

    tfm = LA.HipSpecAugment(time_warp_factor=10, p=0.7)
    seed_all(11)
    tfm.draw(7, 200, 80, None)
    state = (random.random(), int(np.random.randint(1 << 30)), float(torch.rand(1)))
    seed_all(11)
    tfm.draw(7, 200, 80, None)
    assert state == (random.random(), int(np.random.randint(1 << 30)), float(torch.rand(1)))


def test_bicubic_rows_is_torch_interpolate():
    rng = np.random.RandomState(0)
    for in_len, out_len in [(100, 93), (57, 80), (4, 9), (9, 2), (1, 5), (300, 300)]:
        x = rng.randn(in_len, 7).astype(np.float32)
        ref = torch.nn.functional.interpolate(torch.from_numpy(x)[None, None], size=(out_len, 7), mode="bicubic", align_corners=False)[0, 0].numpy()
        got = R.bicubic_rows(x, out_len)
        assert np.abs(got - ref).max() <= 5e-6 * max(1.0, np.abs(ref).max()), (in_len, out_len)


def test_reference_quirks_are_kept():
    # num_frame_masks = 0 ("disable") divides by zero in the reference (signal_transforms.py:252-254); so does the mirror
    tfm = LA.HipSpecAugment(num_frame_masks=0, p=1.0)
    with pytest.raises(ZeroDivisionError):
        tfm.draw(1, 500, 80, None)
    # a sequence so short that the frame-mask width rounds to 0 makes torch.randint(0, 0) fail in both
    with pytest.raises(RuntimeError):
        LA.HipSpecAugment(p=1.0, time_warp_factor=None).draw(1, 5, 80, None)
    sd = LA.HipSpecAugment(time_warp_factor=3, p=0.1).state_dict()
    t2 = LA.HipSpecAugment()
    t2.load_state_dict(sd)
    assert t2.state_dict() == sd and sd["time_warp_factor"] == 3 and sd["p"] == 0.1


def test_global_mvn_restatement_and_module_surface(tmp_path):
    z = np.load(GOLDEN)
    assert np.array_equal(R.global_mvn(z["mvn_in"], z["mvn_means"], z["mvn_stds"]), z["mvn_forward"])
    assert np.array_equal(R.global_mvn(z["mvn_in"], z["mvn_means"], z["mvn_stds"], inverse=True), z["mvn_inverse"])
    mvn = LA.HipGlobalMVN(80)
    mvn.load_state_dict({"norm_means": torch.from_numpy(z["mvn_means"]), "norm_stds": torch.from_numpy(z["mvn_stds"])})
    mvn.to_file(tmp_path / "s.pt")
    again = LA.HipGlobalMVN.from_file(tmp_path / "s.pt")
    assert torch.equal(again.norm_means, mvn.norm_means) and torch.equal(again.norm_stds, mvn.norm_stds) and again.feature_dim == 80
    with pytest.raises(Exception, match="no CPU fallback"):
        mvn(torch.zeros(2, 3, 80))
    with pytest.raises(Exception, match="no CPU fallback"):
        LA.HipSpecAugment()(torch.zeros(2, 300, 80))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_torch_restatement_used_as_timing_baseline_is_the_reference(case):
    from oracle.specaug_torch import TorchSpecAugment

    name, kw, shape, sup, seed = case
    x = make_input(shape, seed)
    seed_all(seed)
    y = TorchSpecAugment(**kw)(torch.from_numpy(x), None if sup is None else torch.tensor(sup, dtype=torch.int32)).numpy()
    assert np.array_equal(y, np.load(GOLDEN)[name])


def test_fast_rng_draws_have_the_reference_distributions():
    """fast_rng=True is an extension: vectorised numpy draws with the same distributions (not the same streams)."""
    np.random.seed(3)
    tfm = LA.HipSpecAugment(time_warp_factor=20, p=0.7, fast_rng=True)
    assert "fast_rng" not in tfm.state_dict()  # the reference's state only
    B, T, F = 400, 600, 80
    seg_rounds, masks = tfm._draw_fast(B, T, F)
    segs = seg_rounds[0]
    applied = np.unique(masks["sequence"])
    assert 0.6 < len(applied) / B < 0.8  # p = 0.7
    assert set(segs["sequence"]).issubset(set(applied.tolist()))
    assert np.all(segs["center"] >= 21) and np.all(segs["center"] < T - 20) and np.all(np.abs(segs["warped"] - segs["center"]) <= 20)
    assert np.all(segs["warped"] != segs["center"])
    fm, tm = masks[masks["axis"] == 2], masks[masks["axis"] == 1]
    assert len(fm) == 2 * len(applied) and len(tm) == min(10, int(np.ceil(0.15 * T / 100))) * len(applied)
    assert np.all(fm["begin"] >= 0) and np.all(fm["end"] <= F) and np.all(fm["end"] - fm["begin"] < 27)
    assert np.all(tm["begin"] >= 0) and np.all(tm["end"] <= T) and np.all(tm["end"] - tm["begin"] < min(100, 0.15 * T // 1))
    # reproducible under numpy's global seed
    np.random.seed(3)
    again = LA.HipSpecAugment(time_warp_factor=20, p=0.7, fast_rng=True)._draw_fast(B, T, F)
    assert np.array_equal(again[1], masks) and np.array_equal(again[0][0], segs)
    # the exact mode is what a default-constructed transform uses
    assert LA.HipSpecAugment().fast_rng is False
