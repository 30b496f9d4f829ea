"""GPU: the on-the-fly mini-batch as a PAIR of launches (hipfeat_speed_bank / hipfeat_minibatch_plan / hipfeat_minibatch_run,
lhotse_amd.augmentation.HipSpeedBank; BASELINE configs[4]) -- mixed-factor speed perturbation + padding rows + descriptor table in one
launch, the collated feature launch behind it.  The bar: BIT-identical to the route of round 3 (one hipfeat_resample launch per factor,
then hipfeat_extract_collated), for both ways the descriptor tables can travel (kernel arguments / staged copy)."""
import numpy as np
import pytest
import torch

import lhotse_amd as LA
from lhotse_amd import _lib
from lhotse_amd import augmentation as A

pytestmark = pytest.mark.gpu
LOG_EPSILON = -23.025850929940457


def _minibatch(seed, n, factors, lo=6000, hi=90000, extra_tail=0):
    rs = np.random.RandomState(seed)
    lens = rs.randint(lo, hi, size=n).astype(np.int64)
    fac = rs.choice(factors, size=n)
    fac[: len(factors)] = factors
    offs = np.concatenate([[0], np.cumsum((lens + 3) & ~3)[:-1]]).astype(np.int64)
    front = int(offs[-1] + lens[-1])
    g = torch.Generator(device="cuda").manual_seed(seed)
    arena = torch.zeros(((front + 3) & ~3) + A.perturbed_tail_floats(lens, fac, 16000) + extra_tail, dtype=torch.float32, device="cuda")
    arena[:front].uniform_(-0.5, 0.5, generator=g)
    return arena, offs, lens, fac, front


def _round3_route(ex, arena, offs, lens, fac, front, want=None, zero_pad=False):
    a = arena.clone()
    po, pl = A.perturb_speed_in_arena(a, offs, lens, fac, 16000, front)
    if want is not None:
        pl = np.minimum(pl, want)
    padded = np.full(len(pl), int(pl.max()), dtype=np.int64) if zero_pad else None
    feats, frames = ex.plan.run_collated(a, po, pl, padded, LOG_EPSILON)
    waves = [a[int(o) : int(o) + int(n)].clone() for o, n in zip(po, pl)]
    return feats, np.asarray(frames), waves


@pytest.mark.parametrize("tables", ["kernel-arguments", "staged"])
@pytest.mark.parametrize("n,factors", [(23, [0.9, 1.0, 1.1]), (7, [1.0]), (5, [1.1]), (180, [0.9, 1.0, 1.1])])
def test_two_launches_equal_the_per_factor_route_bit_for_bit(monkeypatch, tables, n, factors):
    if tables == "staged":
        monkeypatch.setenv("HIPFEAT_MB_NO_INLINE", "1")  # read when the bank is created
    ex = LA.HipFbank()
    arena, offs, lens, fac, front = _minibatch(100 + n, n, factors, hi=30000 if n > 100 else 90000)
    want_feats, want_frames, want_waves = _round3_route(ex, arena, offs, lens, fac, front)
    bank = A.HipSpeedBank(factors, 16000, "cuda")
    before = arena[:front].clone()
    feats, frames, po, pl = bank.extract_collated(ex.plan, arena, offs, lens, bank.index_of(fac), front, LOG_EPSILON)
    torch.cuda.synchronize()
    assert torch.equal(arena[:front], before)  # inputs untouched, unperturbed cuts used in place
    assert np.array_equal(frames, want_frames) and feats.shape == want_feats.shape
    assert torch.equal(feats, want_feats)  # features AND padding rows
    for i in range(n):
        assert torch.equal(arena[int(po[i]) : int(po[i]) + int(pl[i])], want_waves[i])
        if fac[i] == 1.0:
            assert int(po[i]) == int(offs[i]) and int(pl[i]) == int(lens[i])
        else:
            assert int(po[i]) >= front and int(po[i]) % 4 == 0
    # again into a poisoned output (every element must be written by the pair of launches), on a side stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        feats2, frames2, _, _ = bank.extract_collated(ex.plan, arena, offs, lens, bank.index_of(fac), front, LOG_EPSILON)
    s.synchronize()
    assert torch.equal(feats2, want_feats)
    bank.close()


@pytest.mark.parametrize("tables", ["kernel-arguments", "staged"])
def test_several_minibatches_per_launch_pair(monkeypatch, tables):
    """`group_sizes`: K mini-batches (a prefetching loader's) through ONE pair of launches -- every mini-batch its own dense
    (B_k, Tmax_k, F) tensor, bit-identical to K separate calls."""
    if tables == "staged":
        monkeypatch.setenv("HIPFEAT_MB_NO_INLINE", "1")
    ex = LA.HipFbank(LA.HipFbankConfig(edge_rule="batch_zero_pad"))
    sizes = np.array([9, 1, 14, 6], dtype=np.int64) if tables == "kernel-arguments" else np.array([40, 33, 51, 38], dtype=np.int64)
    arena, offs, lens, fac, front = _minibatch(77, int(sizes.sum()), [0.9, 1.0, 1.1], hi=60000)
    bank = A.HipSpeedBank([0.9, 1.1], 16000, "cuda")
    idx = bank.index_of(fac)
    for zero_pad in (False, True):
        outs, frames, po, pl = bank.extract_collated(ex.plan, arena, offs, lens, idx, front, LOG_EPSILON, group_sizes=sizes, zero_pad_batch=zero_pad)
        assert isinstance(outs, list) and len(outs) == len(sizes)
        torch.cuda.synchronize()
        b0 = 0
        for k, n in enumerate(sizes.tolist()):
            sl = slice(b0, b0 + n)
            # the same mini-batch on its own: its cuts are where the grouped call left them (unperturbed in front, perturbed in the tail)
            solo, fr = ex.plan.run_collated(arena, po[sl].copy(), pl[sl].copy(), np.full(n, int(pl[sl].max()), dtype=np.int64) if zero_pad else None, LOG_EPSILON)
            assert outs[k].shape == solo.shape and outs[k].is_contiguous()
            assert np.array_equal(frames[sl], fr) and torch.equal(outs[k], solo)
            b0 += n
    single, f1, _, _ = bank.extract_collated(ex.plan, arena, offs[:9].copy(), lens[:9].copy(), idx[:9].copy(), front, LOG_EPSILON, group_sizes=sizes[:1] * 0 + 9)
    assert isinstance(single, list) and len(single) == 1 and single[0].shape[0] == 9
    with pytest.raises(_lib.HipFeatError, match="the groups hold"):
        bank.extract_collated(ex.plan, arena, offs, lens, idx, front, LOG_EPSILON, group_sizes=sizes[:-1].copy())


def test_truncation_zero_padded_rows_and_mfcc():
    """`max_samples` (lhotse truncates a perturbed cut to the sample count of its manifest, recording.py:1058-1060), the
    edge_rule="batch_zero_pad" framing, and a second kind of plan (MFCC) through the same pair of launches."""
    arena, offs, lens, fac, front = _minibatch(7, 19, [0.9, 1.0, 1.1])
    bank = A.HipSpeedBank([0.9, 1.1], 16000, "cuda")
    idx = bank.index_of(fac)
    out_len = np.array([bank.resamplers[k].output_length(int(n)) if k >= 0 else int(n) for k, n in zip(idx, lens)], dtype=np.int64)
    want = out_len - np.arange(len(lens)) % 3  # 0, 1 or 2 samples to drop
    for ex, zero_pad in ((LA.HipFbank(LA.HipFbankConfig(edge_rule="batch_zero_pad")), True), (LA.HipMfcc(LA.HipMfccConfig(num_filters=40, num_ceps=40)), False)):
        ref_feats, ref_frames, _ = _round3_route(ex, arena, offs, lens, fac, front, want=want, zero_pad=zero_pad)
        feats, frames, po, pl = bank.extract_collated(ex.plan, arena, offs, lens, idx, front, LOG_EPSILON, max_samples=want, zero_pad_batch=zero_pad)
        assert np.array_equal(pl, want) and np.array_equal(frames, ref_frames) and torch.equal(feats, ref_feats)


def test_errors_and_ticket_discipline():
    ex = LA.HipFbank()
    arena, offs, lens, fac, front = _minibatch(9, 12, [0.9, 1.0, 1.1])
    with pytest.raises(_lib.HipFeatError, match="UNSUPPORTED"):
        A.HipSpeedBank([0.95], 16000, "cuda")  # 19:20 is not one of the mixed launch's ratios
    bank = A.HipSpeedBank([0.9, 1.1], 16000, "cuda")
    with pytest.raises(ValueError, match="not in this bank"):
        bank.index_of([0.8])
    idx = bank.index_of(fac)
    with pytest.raises(_lib.HipFeatError, match="arena holds"):
        bank.extract_collated(ex.plan, arena[: front + 64], offs, lens, idx, front, LOG_EPSILON)
    with pytest.raises(_lib.HipFeatError, match="reaches into the arena's tail"):  # (an unperturbed cut beyond tail_start would be overwritten)
        bank.extract_collated(ex.plan, arena, offs, lens, idx, front - 8, LOG_EPSILON)
    short = lens.copy()
    short[3] = 50
    with pytest.raises(_lib.HipFeatError, match="TOO_SHORT"):
        bank.extract_collated(ex.plan, arena, offs, short, idx, front, LOG_EPSILON)
    # a plan whose ticket has been overtaken by 16 newer ones cannot be run any more; an unknown ticket neither
    lib, info = bank.lib, np.zeros(4, dtype=np.int64)
    res = np.empty((3, len(lens)), dtype=np.int64)
    args = (bank.handle, ex.plan.handle, len(lens), _lib.addr(offs), _lib.addr(lens), _lib.addr(idx), None, front, 0, 0, None, _lib.addr(res[0]),
            _lib.addr(res[1]), _lib.addr(res[2]), None, _lib.addr(info))
    lib.check("hipfeat_minibatch_plan", *args)
    first = int(info[0])
    for _ in range(16):
        lib.check("hipfeat_minibatch_plan", *args)
    out = torch.empty((len(lens), int(info[2]), 80), device="cuda")
    for ticket in (first, 10 ** 6):
        with pytest.raises(_lib.HipFeatError, match="not a planned mini-batch"):
            lib.check("hipfeat_minibatch_run", bank.handle, ticket, arena.data_ptr(), arena.numel(), out.data_ptr(), int(info[2]), LOG_EPSILON, 0)
    lib.check("hipfeat_minibatch_run", bank.handle, int(info[0]), arena.data_ptr(), arena.numel(), out.data_ptr(), int(info[2]), LOG_EPSILON,
              int(torch.cuda.current_stream().cuda_stream))
    with pytest.raises(_lib.HipFeatError, match="not a planned mini-batch"):  # a ticket runs once
        lib.check("hipfeat_minibatch_run", bank.handle, int(info[0]), arena.data_ptr(), arena.numel(), out.data_ptr(), int(info[2]), LOG_EPSILON, 0)
    with pytest.raises(_lib.HipFeatError, match="rows per cut"):
        lib.check("hipfeat_minibatch_plan", *args)
        lib.check("hipfeat_minibatch_run", bank.handle, int(info[0]), arena.data_ptr(), arena.numel(), out.data_ptr(), int(info[2]) - 1, LOG_EPSILON, 0)
    # Whisper plans (own normalisation pass) are not served
    with pytest.raises(_lib.HipFeatError, match="UNSUPPORTED"):
        bank.extract_collated(LA.HipWhisperFbank().plan, arena, offs, lens, idx, front, 0.0)
