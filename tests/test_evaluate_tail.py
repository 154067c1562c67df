"""evaluate.py tail: annotations_inverse against the reference's output (golden), result-dict format
(evaluate.py:227-265) by hand-computed expectations, CLI flag surface."""
import numpy as np

from offsetguided_amd import evaluate
from helpers import GOLDEN


def test_annotations_inverse_golden():
    g = np.load(f"{GOLDEN}/eval_tail.npz")
    meta = {'offset': g['offset'], 'scale': g['scale'], 'hflip': False, 'image_id': 1}
    out = evaluate.annotations_inverse(g['poses'], meta)
    assert out.dtype == np.float32 and (out == g['expected']).all()
    assert (g['poses'] == np.load(f"{GOLDEN}/eval_tail.npz")['poses']).all()      # input not mutated


def test_result_dicts():
    poses = np.zeros((2, 17, 6), np.float32)
    poses[0, 0, :3] = [10.126, 20.5, 0.8]
    poses[0, 1, :3] = [0.0, 0.0, 0.0]            # missing joint: stays (0, 0, 0)
    poses[1, :, :3] = [4.0, 6.0, 0.5]
    meta = {'offset': np.array([2.0, -1.0]), 'scale': np.array([2.0, 2.0]), 'hflip': False, 'image_id': 7}
    res, ids = [], []
    evaluate.poses_to_results(poses, meta, res, ids)
    assert ids == [7] and len(res) == 2
    k = res[0]['keypoints']
    # (10.126+2)/2 -> 6.063 -> around(.,2) in float32, then widened (evaluate.py:238-239) ; (20.5-1)/2
    assert len(k) == 51 and k[:3] == [float(np.float32(6.06)), 9.75, 1]
    assert k[3:6] == [1.0, -0.5, 1]                                  # x>0 or y>0 after the inverse shift
    assert abs(res[0]['score'] - 0.8 / 17) < 1e-7 and res[0]['category_id'] == 1
    assert abs(res[1]['score'] - 0.5) < 1e-7
    res2, ids2 = [], []
    evaluate.poses_to_results(np.zeros((0, 17, 6), np.float32), meta, res2, ids2)
    assert res2 == [{'image_id': 7, 'category_id': 1, 'keypoints': [0.0] * 51, 'score': 0.01}]


def test_cli_surface():
    a = evaluate.evaluate_cli(['--topk', '32', '--thre-hmp', '0.04', '--person-thre', '0.04', '--dist-max', '40',
                               '--long-edge', '640', '--batch-size', '8', '--flip-test', '--no-pretrain',
                               '--opt-level', 'O2', '--dataset', 'test-dev'])
    assert a.topk == 32 and a.flip_test and a.all_images and a.headnets == ['hmp', 'omp'] and a.strides == [4, 4]
    assert a.resize_mode == 'bicubic' and a.min_len == 0.5 and a.sort_dim == 2 and a.use_scale is True


def test_soft_nms_matches_reference_golden():
    """decoder/group.py:249-283 (host-side, dead code upstream): golden pose sets from the imported reference."""
    import numpy as np
    from offsetguided_amd.decoder import soft_nms
    from helpers import GOLDEN
    g = np.load(f"{GOLDEN}/soft_nms.npz")
    o = 0
    for n in g["counts"]:
        poses = [p.copy() for p in g["poses_in"][o:o + n]]
        got = soft_nms(poses, suppressed_v=0)
        assert len(got) == n and all(np.array_equal(a, b) for a, b in zip(got, g["poses_out"][o:o + n]))
        o += n
