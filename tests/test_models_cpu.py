"""Model side on the CPU: state_dict layout (968 entries, reference key names), checkpoint round trip with
the reference's dict layout and 'module.' prefixes, BN folding of the inference engine (fp32, no graph)."""
import argparse
import os
import zlib

import numpy as np
import pytest
import torch

from offsetguided_amd import models, synth
from offsetguided_amd.models.seeding import key_seeded_state
from helpers import GOLDEN


def build(seed=0):
    p = argparse.ArgumentParser()
    models.net_cli(p)
    torch.manual_seed(seed)
    model, losses = models.model_factory(p.parse_args(['--no-pretrain']))
    assert [type(l).__name__ for l in losses] == ['HeatMapsLoss', 'OffsetMapsLoss']
    return model


def test_state_dict_layout():
    sd = build().state_dict()
    assert len(sd) == 968
    for key in ('basenet.pre.0.conv.weight', 'basenet.pre.1.skip.0.weight', 'basenet.kps.0.up1.0.conv1.weight',
                'basenet.kps.1.low2.low2.low2.low2.low2.3.bn2.running_var', 'basenet.cnvs.1.bn.num_batches_tracked',
                'basenet.inters.0.conv2.weight', 'basenet.inters_.0.1.bias', 'basenet.cnvs_.0.0.weight',
                'headnets.0.hp_convs.1.bias', 'headnets.1.reg_convs.0.weight'):
        assert key in sd, key
    assert sd['headnets.0.hp_convs.1.weight'].shape == (17, 256, 1, 1)
    assert sd['headnets.1.reg_convs.1.weight'].shape == (38, 256, 1, 1)
    assert sd['basenet.kps.0.low2.low1.0.skip.0.weight'].shape == (384, 256, 1, 1)
    n_params = sum(v.numel() for k, v in sd.items() if 'running' not in k and 'num_batches' not in k)
    assert abs(n_params / 1e6 - 187.73) < 0.01


def test_checkpoint_round_trip(tmp_path):
    a, b = build(1), build(2)
    for m in a.modules():                                   # make the two models really differ
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.normal_(m.weight, 0, 0.05)
    path = os.path.join(tmp_path, 'PoseNet_3_epoch.pth')
    models.save_model(path, 3, 0.25, a)
    ck = torch.load(path, map_location='cpu')
    assert set(ck) == {'epoch', 'train_loss', 'model_state_dict'}
    # reference checkpoints may carry a DataParallel prefix and stray / mis-shaped entries
    ck['model_state_dict'] = {'module.' + k: v for k, v in ck['model_state_dict'].items()}
    ck['model_state_dict']['module.not_in_model.weight'] = torch.zeros(3)
    ck['model_state_dict']['module.headnets.0.hp_convs.0.weight'] = torch.zeros(5, 256, 1, 1)
    torch.save(ck, path)
    keep = b.state_dict()['headnets.0.hp_convs.0.weight'].clone()
    b, opt, epoch, loss, amp = models.load_model(b, path, drop_layers=False, resume_optimizer=False)
    assert (epoch, loss, amp, opt) == (4, 0.25, False, None)
    sa, sb = a.state_dict(), b.state_dict()
    assert torch.equal(sb['headnets.0.hp_convs.0.weight'], keep)          # shape mismatch: kept initialised
    assert all(torch.equal(sa[k], sb[k]) for k in sa if k != 'headnets.0.hp_convs.0.weight')
    with pytest.raises(FileNotFoundError):
        models.load_model(b, os.path.join(tmp_path, 'missing.pth'))       # never blocks on input()


def test_engine_bn_folding_cpu():
    model = build(3)
    for m in model.modules():                                # non-trivial BN statistics and O(1) activations
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
        elif isinstance(m, torch.nn.Conv2d):
            fan = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
            m.weight.data.normal_(0, (1.0 / fan) ** 0.5)
    model.eval()
    x = torch.randn(1, 3, 128, 128)
    with torch.no_grad():
        ref = model(x)
    eng = models.InferenceEngine(model, 1, 128, 128, dtype=torch.float32, device='cpu', use_graph=False)
    out = eng(x)
    for h in (0, 1):
        r, o = ref[h][0][-1], out[h][0][-1]
        assert (o - r).abs().max().item() <= 1e-4 * max(1.0, r.abs().max().item())
    assert out[0][0][0] is None and out[1][1] == [[], []]


def test_backbone_golden_from_reference():
    """tests/golden/backbone128.npz holds the REFERENCE model's head outputs (tools/gen_golden_backbone.py: same
    key-seeded weights in both models, bit-identical there).  Same CPU fp32 path here; the tolerance only covers a
    different host's conv kernels / thread count."""
    g = np.load(os.path.join(GOLDEN, 'backbone128.npz'))
    model = build().eval()
    sd = model.state_dict()
    keys = '\n'.join(f'{k} {tuple(v.shape)}' for k, v in sd.items()).encode()
    assert len(sd) == int(g['n_keys']) and zlib.crc32(keys) == int(g['keys_crc'])     # reference names, order, shapes
    model.load_state_dict(key_seeded_state(sd))
    x = torch.from_numpy(synth.noise_batch(int(g['input_seed']), (1, 3, 128, 128)))
    with torch.no_grad():
        out = model(x)
    for h, name in ((0, 'hm'), (1, 'off')):
        ref = g[name]
        err = np.abs(out[h][0][1].numpy() - ref).max() / np.abs(ref).max()
        assert err <= 1e-5, f'{name}: relative error {err} vs the reference model'
        s0 = out[h][0][0]
        assert abs(s0.std().item() - g[f'{name}_s0_stats'][1]) <= 1e-4 * g[f'{name}_s0_stats'][1]
