"""Composed checks on the GPU (value level, through the drop-in API and the C ABI underneath):

  * BASELINE configs[0]: 1x640x640, torch default init under manual_seed(0) (--initialize-whole False), model -> decode
    against the oracle on the same head outputs (reference recipe: SURVEY 8(d) config 1; models/factory.py:82-125,
    decoder/factory.py:52-96);
  * BASELINE configs[2]: the full bs8 + flip-test batch, merged maps and poses against the oracle on two image pairs
    (decoder/factory.py:98-146);
  * BASELINE configs[3]: two ranks shard a batch and decode with the HIP path (RCCL with two devices, gloo sharing the one
    device otherwise); results gathered on rank 0 equal the oracle's on the whole batch (evaluate.py:139);
  * evaluate.run_images: result dicts equal poses_to_results(oracle.decode(engine outputs)) (evaluate.py:227-265), incl. a
    ragged last batch and --feat-stage 0;
  * the precision statement: poses from the bf16 engine against poses from the eager fp32 module on the same weights and
    a planted signal -- how much of the keypoint set survives the reduced precision (reported and gated)."""
import argparse
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import oracle
from offsetguided_amd import _lib, decoder, models, synth
from offsetguided_amd.config import coco_data as cd
from helpers import FLAGS, assert_poses_match, flip_tables

pytestmark = pytest.mark.gpu
OFLAGS = dict(topk_k=32, thre_hmp=0.04, min_len=0.5, person_thre=0.04, dist_max=40.0)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    _lib.load()
    return torch.device("cuda:0")


def make_args(extra=(), batch=1):
    p = argparse.ArgumentParser()
    models.net_cli(p)
    decoder.decoder_cli(p)
    a = p.parse_args(['--no-pretrain', '--topk', '32', '--thre-hmp', '0.04', '--person-thre', '0.04', '--dist-max', '40',
                      *extra])
    a.batch_size = batch
    return a


def feats_of(hm, off):
    return [([None, hm], [[], []], [[], []]), ([None, off], [[], []], [[], []])]


def test_config1_default_init_model_to_decode(dev):
    """configs[0] composed: seed 0, torch default init, fp32 engine -> HIP decoder == oracle on the same head outputs; the
    bf16 graph engine decodes through the same path."""
    torch.manual_seed(0)
    a = make_args(['--initialize-whole', 'False'])
    model, _ = models.model_factory(a)
    x = torch.randn(1, 3, 640, 640, generator=torch.Generator().manual_seed(0)).to(dev)
    proc = decoder.decoder_factory(a)
    for dtype, graph in ((torch.float32, False), (torch.bfloat16, True)):
        eng = models.InferenceEngine(model, 1, 640, 640, dtype=dtype, device=dev, use_graph=graph)
        out = eng(x)
        hm, off = out[0][0][-1], out[1][0][-1]
        assert hm.shape == (1, 17, 160, 160) and off.shape == (1, 38, 160, 160)
        poses = proc.generate_poses(out)
        ref, _ = oracle.decode(hm.cpu().numpy(), off.cpu().numpy(), cd.COCO_PERSON_SKELETON, **OFLAGS)
        assert_poses_match(ref, poses, 1e-4)
        print(f'config 1 ({dtype}): hm in [{float(hm.min()):.3f}, {float(hm.max()):.3f}], {len(poses[0])} pseudo-poses')
        if dtype == torch.float32:
            assert float(hm.max()) > 0.04 and len(poses[0]) > 0     # the recipe yields candidates above the threshold
    assert model.training and next(model.parameters()).device.type == 'cpu'   # the engine only read the module


def test_flip_full_size_batch(dev):
    """configs[2] at full size: 16 maps in, 8 out; merged maps bit-equal to the oracle's on two image pairs, poses too;
    size-independent properties of the merge on the whole batch."""
    hm, off = synth.synth_batch(21, 8, 640, 640, flip=True)
    assert hm.shape[0] == 16
    proc = decoder.decoder_factory(make_args(batch=8))
    t_hm, t_off = torch.from_numpy(hm).to(dev), torch.from_numpy(off).to(dev)
    mh, _, mo, _, nd = proc.flip_augment(t_hm, [], t_off, [], False, 2)
    assert mh.shape == (8, 17, 160, 160) and mo.shape == (8, 38, 160, 160) and nd == 2
    kp, perm, rev = flip_tables()
    pair = [0, 1, 8, 9]
    rh, ro = oracle.flip_merge(hm[pair], off[pair], kp, perm, rev)
    assert (mh[:2].cpu().numpy() == rh).all() and (mo[:2].cpu().numpy() == ro).all()
    # properties: heatmap merge is the exact mean of the image and its mirrored partner's permuted map
    mirrored = torch.flip(t_hm[8:], [-1])[:, kp]
    assert torch.equal(mh, (t_hm[:8] + mirrored) / 2)
    o5 = mo.view(8, 19, 2, 160, 160)
    assert torch.equal(o5[:, rev], t_off[:8].view(8, 19, 2, 160, 160)[:, rev])      # limbs kept un-averaged (factory.py:138)
    poses = proc.generate_poses(feats_of(t_hm, t_off), flip_test=True)
    ref, _ = oracle.decode(hm[pair], off[pair], cd.COCO_PERSON_SKELETON, flip=(kp, perm, rev), **OFLAGS)
    assert len(poses) == 8
    assert_poses_match(ref, poses[:2], 1e-4)


# ---------------------------------------------------------------------------------- two ranks, HIP decode
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _shard_worker(rank, world, port, two_devices, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch as th
    import torch.distributed as dist
    from offsetguided_amd import sharding
    dev = th.device('cuda', rank if two_devices else 0)
    th.cuda.set_device(dev)
    sharding.init(backend='nccl' if two_devices else 'gloo', device=dev)
    hm, off = synth.synth_batch(77, 6, 256, 256, n_persons=5)                  # the global batch, identical on every rank
    lo, hi = sharding.shard_range(len(hm), rank, world)
    proc = decoder.decoder_factory(make_args(batch=hi - lo))
    mine = proc.generate_poses(feats_of(th.from_numpy(hm[lo:hi]).to(dev), th.from_numpy(off[lo:hi]).to(dev)))
    sharding.barrier(dev)
    slowest = sharding.max_over_ranks(1.0 + rank, dev)
    gathered = sharding.gather_to_rank0(mine)
    ok = True
    if rank == 0:
        ref, _ = oracle.decode(hm, off, cd.COCO_PERSON_SKELETON, **OFLAGS)
        ok = len(gathered) == len(ref)
        for g_, r_ in zip(gathered, ref):
            ok = ok and g_.shape == r_.shape and bool((g_[..., [0, 1, 2, 3, 5]] == r_[..., [0, 1, 2, 3, 5]]).all()) \
                and float(np.abs(g_[..., 4] - r_[..., 4]).max(initial=0)) <= 1e-4
    q.put((rank, ok, slowest, dist.get_backend()))
    dist.destroy_process_group()


def test_two_rank_sharded_decode_hip():
    two = torch.cuda.device_count() >= 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, two, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in results)
    assert all(abs(s - 2.0) < 1e-9 for _, _, s, _ in results)                  # MAX over ranks of (1 + rank)
    assert {b for *_, b in results} == {'nccl' if two else 'gloo'}


# ---------------------------------------------------------------------------------- evaluate.run_images values
class _Loader:
    """Batches in the reference's collate format, from HOST (pageable) tensors, last batch ragged; non-trivial metas."""

    def __init__(self, n_images, batch, size, seed):
        g = torch.Generator().manual_seed(seed)
        self.images = torch.randn(n_images, 3, size, size, generator=g)
        self.metas = [{'image_id': 100 + i, 'offset': np.array([-3.0 * i, 2.0]), 'scale': np.array([0.5 + 0.1 * i, 0.75]),
                       'hflip': False} for i in range(n_images)]
        self.batch = batch

    def __iter__(self):
        for lo in range(0, len(self.images), self.batch):
            hi = min(lo + self.batch, len(self.images))
            yield self.images[lo:hi], [None] * (hi - lo), self.metas[lo:hi]


@pytest.mark.parametrize("stage", [-1, 0])
def test_run_images_values(dev, stage, monkeypatch):
    """run_images' result dicts == poses_to_results(oracle.decode(the head outputs it decoded)), image by image: the maps
    handed to PostProcess.submit are recorded (two engine builds may pick different MIOpen kernels, so the maps are not
    re-computed), the oracle decodes them on the host."""
    from offsetguided_amd import evaluate
    torch.manual_seed(0)
    a = evaluate.evaluate_cli(['--no-pretrain', '--initialize-whole', 'False', '--topk', '32', '--thre-hmp', '0.04',
                               '--person-thre', '0.04', '--dist-max', '40', '--long-edge', '256', '--batch-size', '2',
                               '--print-freq', '1', '--feat-stage', str(stage)])
    model, _ = models.model_factory(a)
    loader = _Loader(5, 2, 256, seed=3)
    seen = []
    build = decoder.decoder_factory

    def recording_factory(args):
        proc = build(args)
        submit = proc.submit

        def spy(features, **kw):
            assert features[0][0][1 - (stage % 2)] is None            # only the decoded stack is computed
            seen.append((features[0][0][stage].cpu().numpy().copy(), features[1][0][stage].cpu().numpy().copy()))
            return submit(features, **kw)
        proc.submit = spy
        return proc
    monkeypatch.setattr(decoder, 'decoder_factory', recording_factory)
    results, ids = evaluate.run_images(a, data_loader=loader, model=model)
    assert ids == [100 + i for i in range(5)] and len(seen) == 3
    exp_results, exp_ids = [], []
    for (hm, off), (images, _, metas) in zip(seen, loader):
        assert hm.shape[0] == 2                                        # the ragged last batch was padded to the engine's
        poses, _ = oracle.decode(hm, off, cd.COCO_PERSON_SKELETON, **OFLAGS)
        for image_poses, meta in zip(poses[:len(metas)], metas):
            evaluate.poses_to_results(image_poses, meta, exp_results, exp_ids)
    assert exp_ids == ids and len(results) == len(exp_results) and len(results) >= 5
    n_real = 0
    for got, exp in zip(results, exp_results):
        assert got['image_id'] == exp['image_id'] and got['category_id'] == 1
        assert got['keypoints'] == exp['keypoints']                       # rounded coordinates + visibility flags
        assert abs(got['score'] - exp['score']) <= 1e-6
        n_real += got['score'] != 0.01
    assert n_real > 0, 'the default-init model should yield pseudo-poses above the thresholds'


@pytest.mark.parametrize("in_flight", [2, 1, 3])
def test_run_images_from_raw_uint8_images(dev, monkeypatch, in_flight):
    """The harness fed with raw uint8 images of mixed sizes: the device input chain (EvalPreprocess) supplies the network
    input and the metas; results map back to ORIGINAL image coordinates through annotations_inverse.  With two batches in flight
    (evaluate.IN_FLIGHT, the default: the second batch of the same shape gets a second engine on the other lane), one, and three."""
    from offsetguided_amd import evaluate, transforms
    monkeypatch.setattr(evaluate, 'IN_FLIGHT', in_flight)
    torch.manual_seed(0)
    a = evaluate.evaluate_cli(['--no-pretrain', '--initialize-whole', 'False', '--topk', '32', '--thre-hmp', '0.04',
                               '--person-thre', '0.04', '--dist-max', '40', '--long-edge', '256', '--batch-size', '2',
                               '--print-freq', '1'])
    model, _ = models.model_factory(a)
    rng = np.random.default_rng(5)
    sizes = [(120, 200), (333, 250), (256, 256), (90, 64)]
    raw = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    loader = [(raw[0:2], [None] * 2, [{'image_id': 1}, {'image_id': 2}]), (raw[2:4], [None] * 2, [{'image_id': 3}, {'image_id': 4}])]
    seen = []
    build = decoder.decoder_factory

    def recording_factory(args):
        proc = build(args)
        submit = proc.submit

        def spy(features, **kw):
            seen.append((features[0][0][-1].cpu().numpy().copy(), features[1][0][-1].cpu().numpy().copy()))
            return submit(features, **kw)
        proc.submit = spy
        return proc
    monkeypatch.setattr(decoder, 'decoder_factory', recording_factory)
    results, ids = evaluate.run_images(a, data_loader=loader, model=model)
    assert ids == [1, 2, 3, 4]
    pre = transforms.EvalPreprocess(256)
    exp_results, exp_ids = [], []
    for (hm, off), (imgs, _, metas) in zip(seen, loader):
        _, pmetas = pre(list(imgs), image_ids=[m['image_id'] for m in metas])
        poses, _ = oracle.decode(hm, off, cd.COCO_PERSON_SKELETON, **OFLAGS)
        for image_poses, meta in zip(poses, pmetas):
            evaluate.poses_to_results(image_poses, meta, exp_results, exp_ids)
    assert exp_ids == ids and len(results) == len(exp_results)
    for got, exp in zip(results, exp_results):
        assert got['image_id'] == exp['image_id'] and got['keypoints'] == exp['keypoints'] and abs(got['score'] - exp['score']) <= 1e-6
    # coordinates are in ORIGINAL image pixels: a detection at the centre of the 256x256 network input maps to the centre of
    # the original image (the pseudo-poses of a random network may also sit in the padding, so no frame check here)
    _, pmetas = pre([raw[1]], image_ids=[2])
    centre = np.zeros((1, 17, 6), np.float32)
    centre[0, :, 0], centre[0, :, 1] = 127.5, 127.5
    back = evaluate.annotations_inverse(centre, pmetas[0])
    assert abs(back[0, 0, 0] - (250 - 1) / 2) < 1.0 and abs(back[0, 0, 1] - (333 - 1) / 2) < 1.0


def test_run_images_fixed_height_engine_cache(dev, monkeypatch):
    """--fixed-height (evaluate.py:150-156, the reference's best published setting: RescaleHighAbsolute + RightDownPad, batch 1,
    the width varies): one engine per padded width, at most ENGINE_CACHE alive (least recently used goes), all of them sharing ONE
    set of folded / tiled weights; results == poses_to_results(oracle.decode(recorded head outputs)) whichever engine ran."""
    from offsetguided_amd import evaluate, transforms
    from offsetguided_amd.models import engine as eng_mod
    torch.manual_seed(0)
    a = evaluate.evaluate_cli(['--no-pretrain', '--initialize-whole', 'False', '--topk', '32', '--thre-hmp', '0.04',
                               '--person-thre', '0.04', '--dist-max', '40', '--long-edge', '256', '--batch-size', '1',
                               '--print-freq', '1', '--fixed-height'])
    model, _ = models.model_factory(a)
    rng = np.random.default_rng(7)
    sizes = [(100, 90), (100, 190), (100, 290), (120, 100), (100, 180), (90, 260), (100, 95)]   # -> widths 256, 512, 768, 256, 512, 768, 256
    raw = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    loader = [([r], [None], [{'image_id': i + 1}]) for i, r in enumerate(raw)]
    seen, built = [], []
    build = decoder.decoder_factory

    def recording_factory(args):
        proc = build(args)
        submit = proc.submit

        def spy(features, **kw):
            seen.append((features[0][0][-1].cpu().numpy().copy(), features[1][0][-1].cpu().numpy().copy()))
            return submit(features, **kw)
        proc.submit = spy
        return proc
    monkeypatch.setattr(decoder, 'decoder_factory', recording_factory)
    engine_cls = models.InferenceEngine

    def counting_engine(*args, **kw):
        e = engine_cls(*args, **kw)
        built.append(e)
        return e
    monkeypatch.setattr(models, 'InferenceEngine', counting_engine)
    monkeypatch.setattr(evaluate, 'ENGINE_CACHE', 2)
    results, ids = evaluate.run_images(a, data_loader=loader, model=model)
    assert ids == [1, 2, 3, 4, 5, 6, 7]
    assert [e.shape[3] for e in built] == [256, 512, 768, 256, 512, 768, 256]      # a cache of two never hits on this cycle of three ...
    assert len({id(e._layers) for e in built}) == 1                                # ... but every build re-uses the same weights
    assert sum(id(e._layers) == id(v[1]) for e in built[:1] for v in eng_mod._layer_cache.values()) == 1
    monkeypatch.setattr(evaluate, 'ENGINE_CACHE', 4)
    del built[:]
    results2, ids2 = evaluate.run_images(a, data_loader=loader, model=model)
    assert len(built) == 3 and ids2 == ids                                          # three shapes, three builds, four hits
    pre = transforms.EvalPreprocess(256, fixed_height=True)
    for res, seen_part in ((results, seen[:7]), (results2, seen[7:])):
        exp_results, exp_ids = [], []
        for (hm, off), (imgs, _, metas) in zip(seen_part, loader):
            _, pmetas = pre(list(imgs), image_ids=[m['image_id'] for m in metas])
            assert hm.shape[2] * 4 == 256 and hm.shape[3] * 4 % 128 == 0
            poses, _ = oracle.decode(hm, off, cd.COCO_PERSON_SKELETON, **OFLAGS)
            for image_poses, meta in zip(poses, pmetas):
                evaluate.poses_to_results(image_poses, meta, exp_results, exp_ids)
        assert exp_ids == ids and len(res) == len(exp_results)
        for got, exp in zip(res, exp_results):
            assert got['image_id'] == exp['image_id'] and got['keypoints'] == exp['keypoints'] and abs(got['score'] - exp['score']) <= 1e-6


def test_shared_layers_follow_the_module_weights(dev):
    """Engines share folded weights only while the module's weights are the same: an in-place update (optimizer step,
    load_state_dict, a write through .data) gives the next engine a fresh fold."""
    from offsetguided_amd.models import engine as eng_mod
    a = make_args(batch=1)
    model, _ = models.model_factory(a)
    e1 = models.InferenceEngine(model, 1, 128, 128, device=dev, use_graph=False)
    e2 = models.InferenceEngine(model, 1, 128, 256, device=dev, use_graph=False)
    assert e1._layers is e2._layers
    x = torch.from_numpy(synth.noise_batch(3, (1, 3, 128, 128))).to(dev)
    before = e1(x)[0][0][-1].clone()
    with torch.no_grad():
        model.headnets[0].hp_convs[-1].weight.data.mul_(2.0)          # .data: no version bump, the checksum catches it
    e3 = models.InferenceEngine(model, 1, 128, 128, device=dev, use_graph=False)
    assert e3._layers is not e1._layers
    after = e3(x)[0][0][-1]
    assert not torch.equal(before, after)
    assert torch.equal(e1(x)[0][0][-1], before)                       # the old engine keeps the weights it was built from


# ---------------------------------------------------------------------------------- precision statement
def _pose_keypoints(poses):
    """{global_idx: score} over the keypoints of all poses of one image."""
    out = {}
    for p in poses:
        for j in range(p.shape[0]):
            if p[j, 5] > 0 or p[j, 2] > 0:
                out[int(p[j, 5])] = float(p[j, 2])
    return out


@pytest.mark.parametrize("dtype,min_common", [(torch.bfloat16, 0.90), (torch.float16, 0.98), (torch.float32, 0.999)])
def test_engine_precision_at_the_pose_level(dev, dtype, min_common):
    """Key-seeded weights + a planted signal (synthetic persons added to the head outputs, as bench.py does): poses decoded
    from the engine's maps vs poses decoded from the eager fp32 module's maps.  Grouping is bit-identical GIVEN identical
    maps; this states how far the maps' precision moves the keypoint set.  Gate: the planted persons' keypoints (scores
    >= 0.3, far above the network's own noise) must all be found at the same pixels."""
    from offsetguided_amd.models.seeding import key_seeded_state
    a = make_args(batch=2)
    model, _ = models.model_factory(a)
    model.load_state_dict(key_seeded_state(model.state_dict()))
    model = model.to(dev).eval()
    x = torch.from_numpy(synth.noise_batch(5, (2, 3, 256, 256))).to(dev)
    hm_s, off_s = synth.synth_batch(9, 2, 256, 256, n_persons=6, hm_noise=0.0, off_noise=0.0)
    hm_s, off_s = torch.from_numpy(hm_s).to(dev), torch.from_numpy(off_s).to(dev)
    with torch.no_grad():
        ref = model(x)
    eng = models.InferenceEngine(model, 2, 256, 256, dtype=dtype, device=dev, use_graph=False)
    out = eng(x)
    scale_hm = 0.02 / float(ref[0][0][-1].abs().max())        # the network's own output as a +-0.02 perturbation ...
    scale_off = 0.5 / float(ref[1][0][-1].abs().max())        # ... and +-0.5 px on the offsets
    proc = decoder.decoder_factory(a)

    def decode(o):
        return proc.generate_poses(feats_of(hm_s + scale_hm * o[0][0][-1].float(), off_s + scale_off * o[1][0][-1].float()))

    p_ref, p_eng = decode(ref), decode(out)
    rel = float((out[0][0][-1] - ref[0][0][-1].float()).abs().max() / ref[0][0][-1].float().abs().max())
    tot = common = strong = strong_common = 0
    dscore = 0.0
    for a_, b_ in zip(p_ref, p_eng):
        ka, kb = _pose_keypoints(a_), _pose_keypoints(b_)
        tot += len(ka)
        both = set(ka) & set(kb)
        common += len(both)
        dscore = max([dscore] + [abs(ka[i] - kb[i]) for i in both])
        s = {i for i, v in ka.items() if v >= 0.3}
        strong += len(s)
        strong_common += len(s & set(kb))
    print(f'{dtype}: head error {rel:.2e} of max; poses {[len(p) for p in p_ref]} vs {[len(p) for p in p_eng]}; '
          f'{common}/{tot} keypoints at identical pixels, planted {strong_common}/{strong}, max |dscore| {dscore:.2e}')
    assert strong > 20 and strong_common >= 0.95 * strong
    assert common >= min_common * tot
    assert dscore <= {torch.bfloat16: 2e-3, torch.float16: 3e-4, torch.float32: 1e-5}[dtype]


def test_own_streams_are_distinct_and_reused_only_after_release(dev):
    """_lib.dedicated_stream / new_stream / release_stream: the process's own HIP streams (torch.cuda.Stream() repeats a pool of 32).
    A key always gets the same stream; new_stream() never hands out a live stream twice -- 40 of them are 40 different handles, none of
    them a lane -- and a released handle is what the next caller gets."""
    from offsetguided_amd import _lib
    lanes = _lib.lane_streams(dev, 2)
    assert [s.cuda_stream for s in lanes] == [s.cuda_stream for s in _lib.lane_streams(dev, 2)] and lanes[0].cuda_stream != lanes[1].cuda_stream
    assert _lib.dedicated_stream(dev, ('k3', 1)).cuda_stream == _lib.dedicated_stream(dev, ('k3', 1)).cuda_stream
    mine = [_lib.new_stream(dev) for _ in range(40)]
    handles = {s.cuda_stream for s in mine}
    assert len(handles) == 40 and not handles & {s.cuda_stream for s in lanes}
    x = torch.ones(1 << 16, device=dev)
    with torch.cuda.stream(mine[7]):
        y = x * 3
    mine[7].synchronize()
    assert float(y.sum()) == 3 * (1 << 16)
    for s in mine[:5]:
        _lib.release_stream(s)
    again = {_lib.new_stream(dev).cuda_stream for _ in range(5)}
    assert again == {s.cuda_stream for s in mine[:5]}
    for s in mine[5:]:
        _lib.release_stream(s)
    for h in again:
        _lib._free_streams[dev.index].add(h)
