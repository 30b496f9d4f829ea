"""GPU: the REAL HIP plan driven exactly the way CutSet.compute_and_store_features_batch drives an extractor
(lhotse/cut/set.py:2365-2404): the main thread calls extract_batch(list of (1, T) CPU tensors) under torch.no_grad() batch
after batch and hands every result to ONE background worker (ThreadPoolExecutor(max_workers=1)) that does
`feat_mat.cpu().numpy()` per cut and "stores" it -- while the main thread is already extracting the next batches into
(possibly recycled) device memory.  lhotse itself is not installed on the GPU box (and /root/reference cannot travel), so the
driver loop is restated here; tests/test_lhotse_dropin.py runs the real drivers against a CPU stand-in of the plan.
Checked: every stored matrix equals the oracle (and the extractor's own single-item result, bit for bit), the frame-count
contract validate_features asserts (lhotse/qa.py:286-311), and that a slow consumer never sees a buffer the producer reused."""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

import lhotse_amd as LA
from _golden import ref32
from oracle.kaldi_ref import RefConfig, RefExtractor

pytestmark = pytest.mark.gpu


def _batches(rng, n_batches, collate):
    for b in range(n_batches):
        n = int(rng.randint(3, 9))
        lens = rng.randint(4000, 48000, size=n)
        waves = [torch.from_numpy((rng.rand(1, int(k)).astype(np.float32) - 0.5)) for k in lens]
        yield b, waves


@pytest.mark.parametrize("device_out", [True, False])
def test_background_save_worker_consumes_device_tensors_while_the_main_thread_extracts(device_out):
    rng = np.random.RandomState(123)
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0"))
    stored = {}

    def save_worker(batch_id, waves, features):  # = _save_worker: per cut .cpu().numpy(), then write + validate
        if batch_id % 3 == 0:
            time.sleep(0.02)  # a slow disk: the producer is several batches ahead
        for i, (w, feat_mat) in enumerate(zip(waves, features)):
            if isinstance(feat_mat, torch.Tensor):
                feat_mat = feat_mat.cpu().numpy()
            assert feat_mat.shape == ((w.shape[-1] + 80) // 160, 80)  # validate_features' frame-count contract
            stored[(batch_id, i)] = feat_mat.copy()

    kept = []
    futures = []
    with ThreadPoolExecutor(max_workers=1) as executor:
        for b, waves in _batches(rng, 24, collate=False):
            with torch.no_grad():
                ins = waves if not device_out else [w.cuda() for w in waves]  # torch in -> torch out, on the input's device
                features = ex.extract_batch(ins, sampling_rate=16000)
            futures.append(executor.submit(save_worker, b, waves, features))
            kept.append((b, waves))
    for f in futures:
        f.result()
    assert len(stored) == sum(len(w) for _, w in kept)
    o32, o64 = ref32(RefConfig(kind="fbank")), RefExtractor(RefConfig(kind="fbank"), np.float64)
    for b, waves in kept:
        for i, w in enumerate(waves):
            x = w[0].numpy()
            got = stored[(b, i)]
            want, truth = o32.extract(x), o64.extract(x)
            rel = np.linalg.norm(got - want) / np.linalg.norm(want)
            assert rel <= 1e-4, (b, i, rel)
            np.testing.assert_array_equal(got, np.asarray(ex.extract(x, 16000)))


def test_collated_form_with_lengths_and_a_background_consumer():
    """collate=True form of the driver: padded (B, Tmax) tensor + int32 lengths (dataset/unsupervised.py:67-84)."""
    rng = np.random.RandomState(7)
    ex = LA.HipFbank(LA.HipFbankConfig(device="cuda:0", edge_rule="batch_zero_pad"))
    o32 = ref32(RefConfig(kind="fbank"))
    results = []

    def save_worker(waves, features):
        results.append((waves, [np.asarray(f.cpu() if isinstance(f, torch.Tensor) else f) for f in features]))

    with ThreadPoolExecutor(max_workers=1) as executor:
        futs = []
        for _ in range(8):
            lens = rng.randint(4000, 32000, size=5)
            waves = [(rng.rand(int(k)).astype(np.float32) - 0.5) for k in lens]
            padded = torch.zeros(len(waves), int(lens.max()))
            for i, w in enumerate(waves):
                padded[i, : len(w)] = torch.from_numpy(w)
            with torch.no_grad():
                feats = ex.extract_batch(padded, sampling_rate=16000, lengths=torch.tensor(lens, dtype=torch.int32))
            futs.append(executor.submit(save_worker, waves, feats))
        for f in futs:
            f.result()
    for waves, feats in results:
        want = o32.extract_batch(waves, "batch_zero_pad")
        for g, w in zip(feats, want):
            assert g.shape == w.shape
            assert np.linalg.norm(g - w) / np.linalg.norm(w) <= 1e-4
