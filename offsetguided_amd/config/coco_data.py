"""COCO keypoint / skeleton tables used by the decoder hot path.

Data tables equal to the reference's config/coco_data.py (keypoint order :56-74,
COCO_PERSON_SKELETON :12-15, the alternative skeletons :17-53, HFLIP :99-116);
the two hflip helpers restate heatmap_hflip (:119-127) and offset_hflip (:130-153).
"""

coco_mean = [0.40789654, 0.44719302, 0.47026115]
coco_std = [0.28863828, 0.27408164, 0.27809835]
data_mean = [0.485, 0.456, 0.406]
data_std = [0.229, 0.224, 0.225]

_SIDES = ('eye', 'ear', 'shoulder', 'elbow', 'wrist', 'hip', 'knee', 'ankle')
COCO_KEYPOINTS = ['nose'] + [f'{lr}_{part}' for part in _SIDES for lr in ('left', 'right')]

# mirror-image partner of every sided keypoint
HFLIP = {f'{a}_{p}': f'{b}_{p}' for p in _SIDES for a, b in (('left', 'right'), ('right', 'left'))}


def _pairs(flat):
    return [(flat[i], flat[i + 1]) for i in range(0, len(flat), 2)]


COCO_PERSON_SKELETON = _pairs([
    0, 1, 0, 2, 1, 2, 1, 3, 2, 4, 5, 6, 4, 6, 3, 5, 5, 7, 7, 9,
    6, 8, 8, 10, 5, 11, 6, 12, 11, 12, 11, 13, 13, 15, 12, 14, 14, 16])

COCO_PERSON_WITH_REDUNDANT_SKELETON = COCO_PERSON_SKELETON + _pairs([
    1, 5, 2, 6, 5, 12, 6, 11, 11, 14, 12, 13, 5, 9, 6, 10, 11, 15, 12, 16, 5, 0, 6, 0])

DENSER_COCO_PERSON_SKELETON = _pairs([
    0, 1, 0, 2, 1, 2, 0, 3, 0, 4, 3, 4, 0, 5, 0, 6, 1, 5, 2, 6, 1, 3, 2, 4, 3, 5, 4, 6, 5, 6,
    5, 11, 6, 12, 5, 12, 6, 11, 11, 12, 5, 7, 6, 8, 7, 9, 8, 10, 5, 9, 6, 10, 7, 8, 9, 10,
    9, 11, 10, 12, 9, 13, 10, 14, 13, 11, 14, 12, 11, 14, 12, 13, 11, 15, 12, 16, 15, 13,
    16, 14, 13, 16, 14, 15, 13, 14, 15, 16])

REDUNDANT_CONNECTIONS = [c for c in DENSER_COCO_PERSON_SKELETON if c not in COCO_PERSON_SKELETON]

KINEMATIC_TREE_SKELETON = _pairs([
    0, 1, 1, 3, 0, 2, 2, 4, 0, 5, 5, 7, 7, 9, 0, 6, 6, 8, 8, 10,
    5, 11, 11, 13, 13, 15, 6, 12, 12, 14, 14, 16])


# per-keypoint OKS sigmas (config/coco_data.py:79-97): nose, eyes, ears, shoulders, elbows, wrists, hips, knees, ankles
COCO_PERSON_SIGMAS = [0.026] + [s for s in (0.025, 0.035, 0.079, 0.072, 0.062, 0.107, 0.087, 0.089) for _ in (0, 1)]


def heatmap_hflip(keypoints, hflip=None):
    """Channel permutation that maps a mirrored heatmap stack back (kp i <- kp perm[i])."""
    table = HFLIP if hflip is None else hflip
    return [keypoints.index(table.get(name, name)) for name in keypoints]


def offset_hflip(keypoints, skeleton, hflip=None):
    """(limb permutation, limbs whose mirror is their own reverse) for mirrored offset maps."""
    table = HFLIP if hflip is None else hflip
    named = [(keypoints[a], keypoints[b]) for a, b in skeleton]
    mirrored = [(table.get(a, a), table.get(b, b)) for a, b in named]
    perm, reverse = list(range(len(skeleton))), []
    for i, (a, b) in enumerate(named):
        if (a, b) in mirrored:
            perm[i] = mirrored.index((a, b))
        if (b, a) in mirrored:
            perm[i] = mirrored.index((b, a))
            reverse.append(i)
    return perm, reverse
