"""Deterministic, key-addressed weights: every state_dict entry is drawn from a generator seeded with crc32 of its
key, so two implementations with the same key names get the same parameters without sharing a checkpoint file
(used to pin this package's Hourglass-104 to the reference's on the same weights)."""
import zlib

import torch


def key_seeded_state(state_dict):
    """state_dict-shaped dict: conv/linear weights ~ N(0, 1/fan_in) (activations stay O(1) through 104 layers),
    BN weight in [0.5, 1.5], BN/conv bias ~ N(0, 0.1), running_mean ~ N(0, 0.1), running_var in [0.5, 1.5]."""
    out = {}
    for key, ref in state_dict.items():
        g = torch.Generator().manual_seed(zlib.crc32(key.encode()))
        if key.endswith('num_batches_tracked'):
            out[key] = torch.zeros_like(ref)
        elif key.endswith('running_var') or (ref.dim() == 1 and key.endswith('weight')):
            out[key] = 0.5 + torch.rand(ref.shape, generator=g)
        elif ref.dim() == 1:
            out[key] = 0.1 * torch.randn(ref.shape, generator=g)
        else:
            fan_in = ref[0].numel()
            out[key] = torch.randn(ref.shape, generator=g) * (1.0 / fan_in) ** 0.5
    return out
