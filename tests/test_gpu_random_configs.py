"""GPU: seeded random sweep over the option surface -- whatever kernel the plan selects (fft512b,
fft256, wave, generic radix-2 / direct DFT) must agree with the float64 oracle to the float32 noise floor of the
reference's own arithmetic (rel-L2 <= max(1e-4, 3 x floor))."""
import warnings

import numpy as np
import pytest

from _golden import record_parity
from _golden import ref32 as ref32_of

import lhotse_amd as LA
from oracle import kaldi_ref as K

pytestmark = pytest.mark.gpu
TABLE = {"fbank": (LA.HipFbank, LA.HipFbankConfig), "mfcc": (LA.HipMfcc, LA.HipMfccConfig), "spectrogram": (LA.HipSpectrogram, LA.HipSpectrogramConfig),
         "log-spectrogram": (LA.HipLogSpectrogram, LA.HipLogSpectrogramConfig)}


from _random_cases import random_case as _random_case  # shared with tests/test_oracle.py (the live-reference pin of ref32)


CASES = [_random_case(np.random.RandomState(1000 + i)) for i in range(160)]


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_random_config_against_float64_oracle(idx):
    kind, cfg = CASES[idx]
    sr = cfg["sampling_rate"]
    rng = np.random.RandomState(idx)
    n_min = int(cfg["frame_length"] * sr) + 8
    lens = [sr, int(2.37 * sr) + 1, max(n_min, sr // 5), 4 * sr]
    xs = [(rng.rand(n).astype(np.float32) - 0.5) * s for n, s in zip(lens, (1.0, 0.05, 0.9, 0.5))]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ex = TABLE[kind][0](TABLE[kind][1](**cfg))
    fields = {k: v for k, v in cfg.items() if k in K.RefConfig.__dataclass_fields__}
    if kind == "mfcc":
        fields.setdefault("num_filters", 23)
    if kind in ("fbank", "mfcc") and K.window_sizes(K.RefConfig(kind=kind, **fields))[2] % 2:
        # an odd fft length has no mel filterbank in the reference either (layers.py:977 asserts)
        with pytest.raises(ValueError, match="must be even"):
            ex.extract_batch(xs, sr)
        return
    ref64 = K.RefExtractor(K.RefConfig(kind=kind, **fields), np.float64)
    ref32 = ref32_of(K.RefConfig(kind=kind, **fields))
    outs = ex.extract_batch(xs, sr)
    assert len(outs) == len(xs)
    for x, got in zip(xs, outs):
        truth = ref64.extract(x)
        want = ref32.extract(x)
        assert got.shape == truth.shape, (ex.kernel_name, kind, cfg, got.shape, truth.shape)
        den = max(np.linalg.norm(truth), 1e-30)
        floor = np.linalg.norm(want - truth) / den
        rel = np.linalg.norm(got - truth) / den
        record_parity("random_configs", (kind, sorted(cfg.items()), len(x)), ex.kernel_name, got, want, truth)
        assert rel <= max(1e-4, 3 * floor), (ex.kernel_name, kind, cfg, len(x), rel, floor)


@pytest.mark.parametrize("seed", range(12))
def test_random_batches_with_the_reference_edge_rule(seed):
    """`edge_rule="batch_zero_pad"` (the reference's _extract_batch: shorter items see zeros past their end)
    on random ragged batches, list and padded-tensor entry points, against the oracle's batch restatement."""
    import torch

    rng = np.random.RandomState(500 + seed)
    kind = ["fbank", "mfcc", "fbank", "log-spectrogram"][seed % 4]
    sr = [16000, 8000, 16000, 24000][seed % 4]
    cfg = dict(sampling_rate=sr, edge_rule="batch_zero_pad")
    lens = sorted((rng.randint(int(0.2 * sr), int(3 * sr), size=rng.randint(2, 7))).tolist(), reverse=bool(seed & 1))
    xs = [(rng.rand(n).astype(np.float32) - 0.5) for n in lens]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ex = TABLE[kind][0](TABLE[kind][1](**cfg))
    fields = {"sampling_rate": sr}
    if kind == "mfcc":
        fields["num_filters"] = 23
    ref = K.RefExtractor(K.RefConfig(kind=kind, **fields), np.float64)
    want = ref.extract_batch(xs, edge_rule="batch_zero_pad")
    got = ex.extract_batch(xs, sr)
    padded = torch.zeros(len(xs), max(lens))
    for i, x in enumerate(xs):
        padded[i, : len(x)] = torch.from_numpy(x)
    got2 = ex.extract_batch(padded, sr, lengths=torch.tensor(lens, dtype=torch.int32))
    for w, g, g2 in zip(want, got, got2):
        assert g.shape == w.shape == g2.shape
        assert np.linalg.norm(g - w) / np.linalg.norm(w) <= 2e-4  # float32 vs float64 truth; log-spectra noise dominated
        assert np.array_equal(np.asarray(g), np.asarray(g2))


@pytest.mark.parametrize("seed", range(16))
def test_random_resampling_ratios(seed):
    from lhotse_amd import augmentation as A
    from oracle import resample_ref as R

    rng = np.random.RandomState(900 + seed)
    rates = [8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000]
    if seed < 8:
        orig, new = rng.choice(rates, size=2, replace=False)
    else:  # speed perturbation factors around 1
        new = 16000
        orig = int(round(16000 * rng.choice([0.9, 0.95, 1.05, 1.1, 0.85, 1.15, 0.97, 1.03])))
    xs = [(rng.rand(n).astype(np.float32) - 0.5) for n in (int(rng.randint(1, 50)), int(rng.randint(1000, 30000)), int(rng.randint(30000, 90000)))]
    r = A.get_or_create_resampler(int(orig), int(new))
    ys = r.resample_batch(xs)
    for x, y in zip(xs, ys):
        want = R.resample(x, int(orig), int(new), dtype=np.float64)
        assert y.numel() == len(want)
        assert np.abs(y.cpu().numpy() - want).max() <= 1e-5, (orig, new, len(x))


@pytest.mark.parametrize("name", ["fbank16k", "mfcc16k", "fbank8k", "mfcc8k", "spec8k", "fbank24k", "whisper"])
def test_repeated_launches_are_bit_identical(name):
    """Race detector: the same ragged batch through the same plan 60 times, from device tensors and from host arrays
    (pinned staging ring, async H2D, descriptor ring), must give bit-identical results every time."""
    import torch

    mk = {
        "fbank16k": (lambda: LA.HipFbank(), 16000),
        "mfcc16k": (lambda: LA.HipMfcc(), 16000),
        "fbank8k": (lambda: LA.HipFbank(LA.HipFbankConfig(sampling_rate=8000, num_filters=40)), 8000),
        "mfcc8k": (lambda: LA.HipMfcc(LA.HipMfccConfig(sampling_rate=8000)), 8000),
        "spec8k": (lambda: LA.HipSpectrogram(LA.HipSpectrogramConfig(sampling_rate=8000)), 8000),
        "fbank24k": (lambda: LA.HipFbank(LA.HipFbankConfig(sampling_rate=24000)), 24000),
        "whisper": (lambda: LA.HipWhisperFbank(), 16000),
    }[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ex = mk[0]()
    sr = mk[1]
    rng = np.random.RandomState(3)
    lens = [int(2.08 * sr) + 30, int(0.98 * sr) - 19, 5 * sr, sr // 3, 3 * sr + 17]
    xs_host = [(rng.rand(n).astype(np.float32) - 0.5) for n in lens]
    xs_dev = [torch.from_numpy(x).cuda() for x in xs_host]
    first = [np.asarray(o).copy() for o in ex.extract_batch(xs_host, sr)]
    for rep in range(60):
        outs = ex.extract_batch(xs_host, sr) if rep % 2 else [o.cpu().numpy() for o in ex.extract_batch(xs_dev, sr)]
        for i, (a, b) in enumerate(zip(outs, first)):
            assert np.array_equal(np.asarray(a), b), (name, rep, i)


@pytest.mark.parametrize("seed", range(6))
def test_random_whisper_and_collated_int16_batches(seed):
    import torch

    from oracle import whisper_ref as W

    rng = np.random.RandomState(700 + seed)
    n_mels = [80, 128, 80, 40, 80, 128][seed]
    lens = [int(v) for v in rng.randint(201, 60000, size=rng.randint(3, 9))] + [201, 160 * 7 + 79, 160 * 7 + 80, 2560 * 3, 2560 * 3 + 1]
    scale = rng.choice([1.0, 0.3, 0.01])
    xs = [((rng.rand(n).astype(np.float32) - 0.5) * scale).astype(np.float32) for n in lens]
    ex = LA.HipWhisperFbank(LA.HipWhisperFbankConfig(num_filters=n_mels))
    filters = W.slaney_mel_filters(16000, 400, n_mels)
    col, flens = ex.extract_collated(xs, 16000)
    assert flens.tolist() == [W.num_rows(n) for n in lens] and col.shape == (len(xs), max(flens.tolist()), n_mels)
    for i, x in enumerate(xs):
        truth = W.log_mel_spectrogram(x, filters, dtype=np.float64)
        got = col[i, : len(truth)].cpu().numpy()
        assert np.abs(got - truth).max() <= 2e-4, (seed, i, len(x), np.abs(got - truth).max())
        assert torch.all(col[i, len(truth):] == np.float32(LA.compat.LOG_EPSILON))
    # int16 PCM through the Kaldi fbank path, collated: bit-identical to the float path
    pcm = [np.round(x / max(scale, 1e-9) * 30000).astype(np.int16) for x in xs if len(x) >= 400]
    fb = LA.HipFbank()
    a, la = fb.extract_collated(pcm, 16000)
    b, lb = fb.extract_collated([p.astype(np.float32) / 32768.0 for p in pcm], 16000)
    assert torch.equal(a, b) and torch.equal(la, lb)


def test_one_extractor_shared_by_several_host_threads():
    """Reader threads of a data pipeline may share one extractor: concurrent extract_batch calls on host arrays (pinned
    staging ring, descriptor ring) must not corrupt each other."""
    import threading

    ex = LA.HipFbank()
    rng = np.random.RandomState(11)
    batches = [[(rng.rand(n).astype(np.float32) - 0.5) for n in rng.randint(8000, 90000, size=5)] for _ in range(6)]
    want = [[np.asarray(o).copy() for o in ex.extract_batch(b, 16000)] for b in batches]
    errors = []

    def worker(tid):
        try:
            for rep in range(25):
                i = (tid + rep) % len(batches)
                outs = ex.extract_batch(batches[i], 16000)
                for a, b in zip(outs, want[i]):
                    if not np.array_equal(np.asarray(a), b):
                        errors.append((tid, rep, i))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:5]
