#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the reference's own known-answer fixture for this path, cut down to a small golden file.

test/fixtures/libri/libri-1088-134315-0000.wav (mono 16 kHz int16, 256 640 samples) with the features the reference stores next to
it (test/fixtures/libri/storage/30c2440c-....npy, 1604 x 40 float32, produced by test/fixtures/libri/recreate.sh with
test/fixtures/libri/fbank40.yml; the values went through lilcom's lossy compression: they agree with the oracle to exactly 2^-6).  Kept here: the first
3 s of PCM and the first 290 feature rows (every frame that lies fully inside those 3 s, so the reflection at the end of the excerpt
does not matter)."""
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = "/root/reference/test/fixtures/libri"


def load_reference_fixture():
    with wave.open(os.path.join(FIX, "libri-1088-134315-0000.wav"), "rb") as f:
        assert f.getframerate() == 16000 and f.getnchannels() == 1 and f.getsampwidth() == 2
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16)
    feats = np.load(os.path.join(FIX, "storage", "30c2440c-93cb-4e83-b382-f2a59b3859b4.npy"))
    return pcm, feats


def main():
    pcm, feats = load_reference_fixture()
    assert pcm.shape == (256640,) and feats.shape == (1604, 40)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "libri_fixture.npz"), pcm=pcm[:48000], feats=feats[:290].astype(np.float32))
    print("wrote", pcm[:48000].shape, feats[:290].shape)
    # round 3 (VERDICT r2 task 4): the WHOLE utterance -- the only real speech the reference holds for this path -- with the reference's own
    # (unquantised) outputs for it, computed here by the reference itself: Fbank(fbank40.yml's 40 filters) and the default 80-filter Fbank
    sys.path.insert(0, ROOT)
    from oracle.make_golden import import_reference

    ex_mod = import_reference()
    x = pcm.astype(np.float32) / 32768.0
    f40 = ex_mod.Fbank(ex_mod.FbankConfig(num_filters=40)).extract(x, 16000)
    f80 = ex_mod.Fbank().extract(x, 16000)
    m13 = ex_mod.Mfcc().extract(x, 16000)
    assert f40.shape == (1604, 40) and f80.shape == (1604, 80) and m13.shape == (1604, 13)
    assert np.abs(f40 - feats).max() <= 2.0 ** -6 + 1e-4  # the stored (lilcom-quantised) fixture
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "libri_full.npz"), pcm=pcm, stored_fbank40=feats.astype(np.float32),
                        fbank40=f40.astype(np.float32), fbank80=f80.astype(np.float32), mfcc13=m13.astype(np.float32))
    print("wrote libri_full.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "libri_full.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
