"""Greedy limb grouping (reference decoder/group.py:17-246) on the device kernel K3."""
import numpy as np
import torch

from .. import _lib
from ..config.coco_data import COCO_KEYPOINTS, COCO_PERSON_SKELETON


class GreedyGroup(object):
    """Groups candidate limbs of ONE image into person skeletons (M, 17, 6) =
    [x, y, v, scale, limb_score, global_idx]; constructor as decoder/group.py:29-37.

    `group_skeletons` keeps the reference's numpy-in / numpy-out contract;
    `group_batch` is the device-resident form used by PostProcess (one workgroup per image,
    one D2H copy of the finished poses)."""

    MMAX = 128  # rows of the LDS-resident partial-skeleton table; grown on overflow

    def __init__(self, person_thre, *, sort_dim=2, dist_max=10, use_scale=False,
                 keypoints=COCO_KEYPOINTS, skeleton=COCO_PERSON_SKELETON):
        self.person_thre = person_thre
        self.use_scale = use_scale
        self.sort_dim = sort_dim
        self.skeleton = skeleton
        self.keypoints = keypoints
        self.dist_max = dist_max
        self.n_keypoints = len(keypoints)

    def group_skeletons(self, limbs):
        assert len(limbs) == len(self.skeleton), 'check the skeleton config and input limbs Tensor'
        if not torch.cuda.is_available():
            raise _lib.OgError('GreedyGroup.group_skeletons needs a HIP device (no CPU path)')
        dev = torch.device('cuda', torch.cuda.current_device())
        t = torch.from_numpy(np.ascontiguousarray(limbs, dtype=np.float32)).to(dev)
        return self.group_batch(t.unsqueeze(0))[0]

    def group_batch(self, limbs):
        """limbs: device tensor (N, L, K, 13) -> list of N float32 arrays (M_i, n_kp, 6).

        One D2H copy of (counts, status) and one of the finished poses; this is the only host
        synchronisation of the decoder (the reference syncs at .cpu().numpy() before grouping)."""
        mmax = self.MMAX
        limit = limbs.shape[1] * limbs.shape[2]
        while True:
            poses, meta = self.group_device(limbs, mmax)
            meta_h = meta.cpu().numpy()                      # [counts | status]
            n = limbs.shape[0]
            counts, status = meta_h[:n], meta_h[n:]
            if not status.any() or mmax >= limit:
                break
            mmax = min(mmax * 4, limit)                      # table overflow is rare: retry larger
        host = poses[:, :max(int(counts.max()), 1)].cpu().numpy()
        return [host[i, :counts[i]].copy() for i in range(n)]

    def group_device(self, limbs, mmax=None):
        """Stream-ordered, no host sync: returns (poses (N, mmax, n_kp, 6), meta int32 (2N,) =
        [poses per image | overflow status per image]) device tensors."""
        limbs = _lib.require_device(limbs, 'limbs')
        n, n_limbs, k, width = limbs.shape
        assert n_limbs == len(self.skeleton) and width == 13, 'check the skeleton config and input limbs Tensor'
        dev = limbs.device
        lib = _lib.load()
        mmax = mmax or self.MMAX
        jf = _lib.int_table([a for a, _ in self.skeleton], dev)
        jt = _lib.int_table([b for _, b in self.skeleton], dev)
        poses = torch.empty((n, mmax, self.n_keypoints, 6), dtype=torch.float32, device=dev)
        meta = torch.empty(2 * n, dtype=torch.int32, device=dev)
        nbytes = lib.og_group_workspace_bytes(n, n_limbs, k, self.n_keypoints, mmax)
        ws = _lib.workspace(dev, nbytes, 'group')
        with _lib.stage_timer('k3_group', dev):
            _lib.check(lib.og_greedy_group_f32(
                _lib.ptr(limbs), n, n_limbs, k, _lib.ptr(jf), _lib.ptr(jt), self.n_keypoints,
                float(self.person_thre), float(self.dist_max), int(bool(self.use_scale)), int(self.sort_dim),
                mmax, _lib.ptr(poses), _lib.ptr(meta), _lib.ptr(meta[n:]), _lib.ptr(ws), ws.numel(),
                _lib.stream_ptr(dev)), lib)
        return poses, meta


def soft_nms(subset, suppressed_v=0):
    """Keypoint-level suppression over finished poses (decoder/group.py:249-289 there; dead code upstream: its only call,
    :183, is commented out).  Host-side, on the poses already copied back; modifies `subset` in place and returns it.

    Every keypoint type has its own occupancy plane.  Poses are visited in order: a keypoint whose (clipped, truncated)
    pixel is already covered gets the score `suppressed_v`; otherwise it covers the square of half-width
    max(10, keypoint scale) round itself (at least one pixel, clipped to the plane).  The planes are independent, so this
    walks them one keypoint type at a time."""
    if not len(subset):
        return subset
    height = int(max(float(pose[:, 1].max()) for pose in subset) + 1)
    width = int(max(float(pose[:, 0].max()) for pose in subset) + 1)
    n_types = len(subset[0])
    for j in range(n_types):
        covered = np.zeros((height, width), dtype=bool)
        for pose in subset:
            assert len(pose) == n_types
            x, y, v = pose[j, 0], pose[j, 1], pose[j, 2]
            if v == -1:
                continue
            col = int(min(max(x, 0.0), width - 1))
            row = int(min(max(y, 0.0), height - 1))
            if covered[row, col]:
                pose[j, 2] = suppressed_v
                continue
            half = max(10.0, pose[j, 3])
            x0, y0 = max(0, int(x - half)), max(0, int(y - half))
            x1 = max(x0 + 1, min(width, int(x + half) + 1))
            y1 = max(y0 + 1, min(height, int(y + half) + 1))
            covered[y0:y1, x0:x1] = True
    return subset
