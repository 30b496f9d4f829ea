"""GPU: mixed-factor speed perturbation of a packed mini-batch inside ONE arena buffer (lhotse_amd.augmentation.perturb_speed_in_arena,
the device-resident form of PerturbSpeed + Speed: lhotse/dataset/cut_transforms/perturb_speed.py:8-47, lhotse/augmentation/torchaudio.py:26-42)
followed by the collated Fbank launch -- BASELINE configs[4] as bench.py --config onthefly runs it."""
import numpy as np
import pytest
import torch

import lhotse_amd as LA
from lhotse_amd import augmentation as A
from oracle import resample_ref as R
from _golden import ref32
from oracle.kaldi_ref import RefConfig, RefExtractor

pytestmark = pytest.mark.gpu
LOG_EPSILON = -23.025850929940457


def test_arena_perturbation_equals_per_cut_speed_then_fbank():
    rs = np.random.RandomState(3)
    lens = rs.randint(8000, 70000, size=23).astype(np.int64)
    fac = rs.choice([0.9, 1.0, 1.1], size=len(lens))
    fac[:3] = [0.9, 1.0, 1.1]
    waves = [(rs.rand(int(n)).astype(np.float32) - 0.5) for n in lens]
    offs = np.concatenate([[0], np.cumsum((lens + 3) & ~3)[:-1]]).astype(np.int64)
    front = int(offs[-1] + lens[-1])
    arena = torch.zeros(((front + 3) & ~3) + A.perturbed_tail_floats(lens, fac, 16000), dtype=torch.float32, device="cuda")
    for w, o in zip(waves, offs):
        arena[int(o) : int(o) + len(w)] = torch.from_numpy(w).cuda()
    before = arena[:front].clone()
    po, pl = A.perturb_speed_in_arena(arena, offs, lens, fac, 16000, front)
    assert torch.equal(arena[:front], before)  # the inputs are untouched: unperturbed cuts are used in place
    ex = LA.HipFbank()
    feats, flens = ex.plan.run_collated(arena, po, pl, None, LOG_EPSILON)
    o32 = ref32(RefConfig(kind="fbank"))
    for i, w in enumerate(waves):
        y = R.speed(w, 16000, float(fac[i])) if fac[i] != 1.0 else w
        assert int(pl[i]) == len(y)
        got_wave = arena[int(po[i]) : int(po[i]) + int(pl[i])].cpu().numpy()
        assert np.abs(got_wave - y).max() <= 1e-5  # the resampler against the oracle (bit-identical to HipSpeed: same kernel)
        if fac[i] == 1.0:
            assert int(po[i]) == int(offs[i]) and np.array_equal(got_wave, w)
        want = o32.extract(y)
        got = feats[i, : int(flens[i])].cpu().numpy()
        assert got.shape == want.shape and np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-4
        assert bool((feats[i, int(flens[i]) :] == LOG_EPSILON).all())
    # a second pass over the same arena gives the same bits (the bench repeats it every step)
    po2, pl2 = A.perturb_speed_in_arena(arena, offs, lens, fac, 16000, front)
    feats2, _ = ex.plan.run_collated(arena, po2, pl2, None, LOG_EPSILON)
    assert np.array_equal(po, po2) and torch.equal(feats, feats2)
    with pytest.raises(ValueError, match="arena too small"):
        A.perturb_speed_in_arena(arena[: front + 100], offs, lens, fac, 16000, front)
