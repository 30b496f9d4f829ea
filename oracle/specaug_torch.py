"""
TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

The reference's SpecAugment as the sequence of torch calls it makes per sequence (lhotse/dataset/signal_transforms.py:
173-371: clone, per-sequence p-check, two bicubic F.interpolate + cat, mean, slice fills), written against plain torch so
that it can run on the GPU box (where /root/reference does not exist) as the timing baseline of tools/bench_specaug.py.
tests/test_specaug_oracle.py checks it against the reference-generated goldens for the same seeds.
"""
import math
import random

import numpy as np
import torch


class TorchSpecAugment(torch.nn.Module):
    def __init__(self, time_warp_factor=80, num_feature_masks=2, features_mask_size=27, num_frame_masks=10, frames_mask_size=100,
                 max_frames_mask_fraction=0.15, p=0.9):
        super().__init__()
        self.time_warp_factor, self.num_feature_masks, self.features_mask_size = time_warp_factor, num_feature_masks, features_mask_size
        self.num_frame_masks, self.frames_mask_size, self.max_frames_mask_fraction, self.p = num_frame_masks, frames_mask_size, max_frames_mask_fraction, p

    def forward(self, features, supervision_segments=None):
        features = features.clone()
        if supervision_segments is None:
            for i in range(features.size(0)):
                features[i] = self._single(features[i])
        else:
            for i, start, n in supervision_segments:
                features[i, start : start + n] = self._single(features[i, start : start + n], warp=True, mask=False)
            for i in range(features.size(0)):
                features[i] = self._single(features[i], warp=False, mask=True)
        return features

    def _single(self, features, warp=True, mask=True):
        if random.random() > self.p:
            return features
        if warp and self.time_warp_factor is not None and self.time_warp_factor >= 1:
            features = _time_warp(features, self.time_warp_factor)
        if mask:
            mean = features.mean()
            features = _mask(features, self.features_mask_size, self.num_feature_masks, mean, 2)
            tot = self.max_frames_mask_fraction * features.size(0)
            n = min(self.num_frame_masks, math.ceil(tot / self.frames_mask_size))
            features = _mask(features, min(self.frames_mask_size, tot // n), n, mean, 1)
        return features


def _mask(features, mask_size, mask_times, mask_value, axis):
    features = features.unsqueeze(0)
    values = torch.randint(int(0), int(mask_size), (1, mask_times))
    min_values = torch.rand(1, mask_times) * (features.size(axis) - values)
    starts, ends = min_values.long().squeeze(), (min_values.long() + values.long()).squeeze()
    if mask_times == 1:
        starts, ends = [starts], [ends]
    for a, b in zip(starts, ends):
        if axis == 1:
            features[:, a:b] = mask_value
        else:
            features[:, :, a:b] = mask_value
    return features.squeeze(0)


def _time_warp(features, factor):
    t = features.size(0)
    if t - factor <= factor + 1:
        return features
    center = np.random.randint(factor + 1, t - factor)
    warped = np.random.randint(center - factor, center + factor + 1)
    if warped == center:
        return features
    f = features.unsqueeze(0).unsqueeze(0)
    left = torch.nn.functional.interpolate(f[:, :, :center, :], size=(warped, f.size(3)), mode="bicubic", align_corners=False)
    right = torch.nn.functional.interpolate(f[:, :, center:, :], size=(t - warped, f.size(3)), mode="bicubic", align_corners=False)
    return torch.cat((left, right), dim=2).squeeze(0).squeeze(0)
