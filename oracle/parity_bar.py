"""
TEST INFRASTRUCTURE (checker only; never imported by the product): the ONE parity statement of this repository for the
headline workload (log-mel filterbank), used verbatim by `bench.py`'s in-run parity leg, by
`tests/test_gpu_parity.py::test_headline_parity_multi_seed` and quoted in DESIGN.md section 2.

`north_star` asks for "within 1e-4 relative float tolerance" of the reference.  For a float32 pipeline that ends in a
logarithm this is read as three clauses, all of which must hold on the pooled values of a comparison:

  (1) norm-wise:       ||hip - ref32||_2 / ||ref32||_2 <= 1e-4                             per cut
  (2) linear domain, against float64 truth, for EVERY value:
                       |exp(hip) - exp(f64)| <= L * (1e-4 * exp(f64) + eps),   L = max(1, K_LIN * worst share of ref32),  K_LIN = 1.25
                       with eps = 1.1920929e-07, the constant at which the reference itself clamps every mel energy
                       (layers.py:536-538, 572: `max(mel, eps).log()`), i.e. what it treats as nothing, and "worst share of
                       ref32" = max |exp(ref32) - exp(f64)| / (1e-4 * exp(f64) + eps) over the same cuts: the mel ENERGY of the HIP
                       kernel is within 1e-4 relative (+ the reference's own floor) of the true energy, or at most K times as far
                       out as the reference's own worst energy.  Rounds 4 and early 5 stated this clause as hip-vs-ref32 with L = 1; the
                       reference itself does not meet that against float64 (measured WITHOUT the kernel, /root/reference on 32 x 10 s of
                       noise: worst share 1.048, 3 values of 2.56 M over 1), so two float32 pipelines that are each at their rounding
                       floor can differ by more than the tolerance -- bench.py's second rank (seed 1235) showed exactly that: ONE value of
                       10.2 M at 1.068 (profiles/r05_bench_2ranks_gloo_first_attempt.txt).  The hip-vs-ref32 figure is still reported
                       (`lin_margin_max`, `lin_bad`).
  (3) element-wise, log domain, against float64 truth:
                       max|hip - f64| <= max(2e-3, K * max|ref32 - f64|),   K = 3
                       i.e. the HIP kernel's worst value is at most 3 times as far from the float64 result as the reference's
                       own worst value on the same cuts (or inside the flat 2e-3 bar of the goldens) -- the same
                       `max(2e-3, 3 x floor)` escape the golden suite has used since round 1.

ref32 (round 5): the reference's REAL float32 arithmetic -- oracle/kaldi_torch.reference_f32(), the reference's own
torch call sequence (torch.fft.rfft on float32 frames, layers.py:32-42), array_equal to the live reference on full-size cuts.
Rounds 1-4 took ref32 from oracle/kaldi_ref.py's float32 mode, whose FFT is numpy's float64 rfft rounded to complex64: its floor
max|ref32 - f64| (4-5e-4 on 10 s of noise) understated the reference's own error (1.3e-3 ... 2.1e-3) about 4x, which is where round 4's
"K = 2 ... 7.6, K_allowed = 10" came from.  K went back to 3 in the same change that swapped the floor, BEFORE the re-measurement
(a tolerance is not to be fitted to the implementation's error); `figures(..., alt32=)` carries the numpy floor
next to the real one so that profiles/r05_parity.json shows both.

What (3) is about (profiles/r03_parity_probe.txt): on 10 s of uniform noise a few values in a million are mel energies within
~4 nats of the clamp (1e-7 of the row's median, after pre-emphasis has pushed the low bins 36 dB under the near-Nyquist ones).
There every float32 pipeline sits at its rounding floor; the reference's own tolerance precedent is decimal=3
(test/features/test_kaldifeat_features.py:103-116).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

REL_L2_TOL = 1e-4
LIN_RTOL = 1e-4
LIN_ATOL = 1.1920929e-07  # the reference's mel floor (layers.py:536)
ABS_TOL = 2e-3
K_FLOOR = 3.0   # clause 3 (element-wise, log domain); measured 0.53 ... 2.30 on the twelve inputs of round 5
K_LIN = 1.25    # clause 2 (linear domain); measured hip / reference share 0.99 ... 1.01 on the same inputs (VERDICT r5 task 3: 3 was slack)
# FROZEN (round 6).  The statement below changes again only together with a REFERENCE-side measurement committed under profiles/
# (i.e. evidence about the reference's own float32 floor, never a measurement of the HIP kernel) -- DESIGN.md section 2.
STATEMENT_VERSION = "r6-frozen-1"


def figures(got: np.ndarray, want: np.ndarray, truth: np.ndarray, log_mel: bool = True, alt32: np.ndarray = None) -> Dict:
    """Error figures of ONE cut: hip (`got`) vs ref32 (`want` = the reference's float32 arithmetic, oracle/kaldi_torch.py) and both
    against the float64 oracle (`truth`).  `alt32` (optional) = oracle/kaldi_ref.py's float32 mode (float64 FFT rounded to float32,
    the ref32 of rounds 1-4): its floor and the kernel's distance to it are recorded next to the real ones."""
    got64, want64 = got.astype(np.float64), want.astype(np.float64)
    d = np.abs(got64 - want64)
    fl = np.abs(want64 - truth)
    own = np.abs(got64 - truth)
    lin_margin, lin_bad, lin_own, lin_floor, lin_own_over, lin_floor_over = 0.0, 0, 0.0, 0.0, 0, 0
    if log_mel:
        ew, eg, et = np.exp(want64), np.exp(got64), np.exp(truth)
        m = np.abs(eg - ew) / (LIN_RTOL * ew + LIN_ATOL)  # hip vs ref32 (reported)
        lin_margin, lin_bad = float(m.max()), int((m > 1.0).sum())
        tol = LIN_RTOL * et + LIN_ATOL
        mo, mf = np.abs(eg - et) / tol, np.abs(ew - et) / tol  # hip vs truth, ref32 vs truth (clause 2)
        lin_own, lin_floor, lin_own_over, lin_floor_over = float(mo.max()), float(mf.max()), int((mo > 1.0).sum()), int((mf > 1.0).sum())
    over = d > ABS_TOL
    alt = {}
    if alt32 is not None:
        alt64 = alt32.astype(np.float64)
        alt = {"alt_floor_abs": float(np.abs(alt64 - truth).max()), "alt_abs": float(np.abs(got64 - alt64).max()),
               "alt_vs_ref32_abs": float(np.abs(alt64 - want64).max())}
    return {
        **alt,
        "rel": float(np.linalg.norm(got64 - want64) / np.linalg.norm(want64)),
        "abs": float(d.max()),
        "within": int((d <= 1e-3 + 1e-4 * np.abs(want64)).sum()),
        "total": int(d.size),
        "floor_rel": float(np.linalg.norm(want64 - truth) / np.linalg.norm(truth)),
        "floor_abs": float(fl.max()),
        "own_abs": float(own.max()),
        "floor_sq": float((fl ** 2).sum()),
        "own_sq": float((own ** 2).sum()),
        "lin_bad": lin_bad,
        "lin_margin": lin_margin,
        "lin_own": lin_own,
        "lin_floor": lin_floor,
        "lin_own_over": lin_own_over,
        "lin_floor_over": lin_floor_over,
        "over": int(over.sum()),
        "over_ref_max": float(want[over].max()) if bool(over.any()) else None,
    }


def fold(stats: List[Dict]) -> Dict:
    """Pool the per-cut figures of one comparison (one rank's sample)."""
    n = max(1, sum(s["total"] for s in stats))
    alt = {}
    if stats and all("alt_floor_abs" in s for s in stats):
        alt = {"numpy32_vs_f64_max_abs": max(s["alt_floor_abs"] for s in stats), "hip_vs_numpy32_max_abs": max(s["alt_abs"] for s in stats),
               "numpy32_vs_ref32_max_abs": max(s["alt_vs_ref32_abs"] for s in stats)}
    return {
        **alt,
        "rel_l2_max": max(s["rel"] for s in stats),
        "max_abs_max": max(s["abs"] for s in stats),
        "frac_within": sum(s["within"] for s in stats) / n,
        "n": len(stats),
        "oracle_f32_vs_f64_rel_l2_max": max(s["floor_rel"] for s in stats),
        "oracle_f32_vs_f64_max_abs": max(s["floor_abs"] for s in stats),
        "hip_vs_f64_max_abs": max(s["own_abs"] for s in stats),
        "oracle_f32_vs_f64_rms": (sum(s["floor_sq"] for s in stats) / n) ** 0.5,
        "hip_vs_f64_rms": (sum(s["own_sq"] for s in stats) / n) ** 0.5,
        "lin_bad": sum(max(s["lin_bad"], 0) for s in stats),
        "lin_margin_max": max(s["lin_margin"] for s in stats),
        "lin_own_max": max(s["lin_own"] for s in stats),
        "lin_floor_max": max(s["lin_floor"] for s in stats),
        "lin_own_over1": sum(s["lin_own_over"] for s in stats),
        "lin_floor_over1": sum(s["lin_floor_over"] for s in stats),
        "n_over_2e-3": sum(s["over"] for s in stats),
        "over_ref_value_max": max([s["over_ref_max"] for s in stats if s["over_ref_max"] is not None], default=None),
        "n_values": n,
    }


def verdict(f: Dict) -> Dict:
    """The three clauses on folded figures (`fold` output, or the max-reduction of several ranks' folds)."""
    k = f["hip_vs_f64_max_abs"] / max(f["oracle_f32_vs_f64_max_abs"], 1e-30)
    bar = max(ABS_TOL, K_FLOOR * f["oracle_f32_vs_f64_max_abs"])
    # keys are indexed directly: a folded / rank-reduced dict that lost a figure must raise, not pass
    lin_bar = max(1.0, K_LIN * f["lin_floor_max"])
    v = {
        "statement_version": STATEMENT_VERSION,
        "pass_rel_l2": bool(f["rel_l2_max"] <= REL_L2_TOL),
        "pass_linear": bool(f["lin_own_max"] <= lin_bar),
        "linear_bar_share_of_tolerance": float(lin_bar),
        "pass_elementwise": bool(f["hip_vs_f64_max_abs"] <= bar),
        "elementwise_bar": float(bar),
        "K_measured": float(k),
        "K_allowed": K_FLOOR,
        "K_linear_allowed": K_LIN,
        "K_linear_measured": float(f["lin_own_max"] / max(f["lin_floor_max"], 1e-30)),
        # the hip-vs-ref32 form of clause 2 (L = 1, rounds 4 / early 5), reported next to the verdict: values outside and the worst share
        "linear_hip_vs_ref32_outside": int(f["lin_bad"]),
        "linear_hip_vs_ref32_worst_share": float(f["lin_margin_max"]),
    }
    v["pass"] = v["pass_rel_l2"] and v["pass_linear"] and v["pass_elementwise"]
    if "numpy32_vs_f64_max_abs" in f:  # the floor rounds 1-4 used, side by side (not part of the verdict)
        v["K_against_numpy32_floor"] = float(f["hip_vs_f64_max_abs"] / max(f["numpy32_vs_f64_max_abs"], 1e-30))
    return v


STATEMENT = (f"[{STATEMENT_VERSION}] pass = per-cut rel_l2(hip, ref32) <= 1e-4  AND  every value: |exp(hip) - exp(f64)| <= L (1e-4 exp(f64) + 1.19e-7 [the reference's own mel "
             "floor]), L = max(1, 1.25 x the worst such share of ref32 itself)  AND  max|hip - f64| <= max(2e-3, 3 x max|ref32 - f64|); ref32 = the "
             "reference's own float32 torch call sequence (oracle/kaldi_torch.py, array_equal to the live reference), f64 = the same algorithm in float64 "
             "(oracle/kaldi_ref.py); oracle/parity_bar.py, enforced on the same inputs by tests/test_gpu_parity.py::test_headline_parity_multi_seed")
