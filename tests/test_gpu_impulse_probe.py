"""GPU: impulse probe of every kernel family's pre-emphasis and framing.  One impulse per frame at every position p of the
frame (rectangular window, no DC removal, snip_edges): the power spectrum of such a frame is |1 - c e^{-jw}|^2 whatever p is, so
its autocorrelation has r[0] = 1 + c^2 and r[1] = -c (and, for the last sample of the frame, r[0] = 1: the tap falls outside).
A tap that lands on the wrong sample -- e.g. at the register boundaries of a lane-per-sample layout -- shows up at another lag.
(This is the probe that localised such a bug while the wave kernel was reworked)"""
import warnings

import numpy as np
import pytest

import lhotse_amd as LA

pytestmark = pytest.mark.gpu

CASES = [  # sampling rate, extra config, kernel family expected
    (8000, {}, "fft256_kernel"),
    (16000, {}, "fft512b_kernel"),
    (16000, {"use_energy": True}, "wave_kernel<4>"),
    (22050, {}, "wave_kernel<8>"),
    (44100, {}, "wave_kernel<16>"),
    (16000, {"round_to_power_of_two": False}, "generic"),
]


@pytest.mark.parametrize("sr,kw,kernel", CASES, ids=[c[2] for c in CASES])
def test_every_impulse_position_gets_its_tap_on_the_next_sample(sr, kw, kernel):
    c = 0.97
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ex = LA.HipSpectrogram(LA.HipSpectrogramConfig(sampling_rate=sr, remove_dc_offset=False, window_type="rectangular", snip_edges=True,
                                                       preemph_coeff=c, **kw))
    assert kernel in ex.kernel_name, ex.kernel_name
    n, shift, fft = ex.plan.n, ex.plan.shift, ex.plan.fft
    period = 2 * n + 1  # > n: at most one impulse per frame; coprime with the shift: every position is reached
    while np.gcd(period, shift) != 1:
        period += 1
    need = shift * period + 4 * n
    x = np.zeros(need, dtype=np.float32)
    pos = np.arange(n, len(x) - n, period)
    x[pos] = 1.0
    y = ex.extract(x, sr).astype(np.float64)  # (T, fft/2 + 1) power
    if kw.get("use_energy"):
        y = y[:, 1:]  # bin 0 holds the log-energy: drop DC (the autocorrelation below then misses a constant only)
    seen = {}
    for t in range(y.shape[0]):
        lo = t * shift
        inside = pos[(pos >= lo) & (pos < lo + n)]
        if len(inside) != 1:
            continue
        p = int(inside[0] - lo)
        if p in seen:
            continue
        if kw.get("use_energy"):
            full = np.concatenate([[0.0], y[t], y[t][-2::-1]])
        else:
            full = np.concatenate([y[t], y[t][-2:0:-1]])
        r = np.fft.ifft(full).real
        if kw.get("use_energy"):
            r = r - r[fft // 2]  # remove the constant introduced by zeroing DC (lag fft/2 carries no tap)
        seen[p] = (r[0], r[1], np.abs(r[2 : fft // 2 - 1]).max())
    assert len(seen) == n, (len(seen), n)
    for p, (r0, r1, rest) in seen.items():
        if p == 0:  # y[0] = x[0] - c x[0]
            want0, want1 = (1 - c) ** 2 + c * c, -c * (1 - c)
        elif p == n - 1:  # the tap falls outside the frame
            want0, want1 = 1.0, 0.0
        else:
            want0, want1 = 1 + c * c, -c
        assert abs(r0 - want0) < 2e-3 and abs(r1 - want1) < 2e-3 and rest < 2e-3, (p, r0, r1, rest)
