"""Portable synthetic head outputs for the decoder path.

Stick-figure "persons" are rendered into stride-4 keypoint heatmaps and guiding-offset
maps with the reference encoder's conventions (cell centre grid ``i*stride + stride/2 - 0.5``,
encoder/heatmap.py:122-123; Gaussian sigma 7 px, encoder/heatmap.py:20; offsets in input-pixel
units inside a patch round the from-joint, encoder/offset.py:154-197), plus noise.

All randomness comes from a counter-based integer hash (splitmix64), not from
``numpy.random``: the streams are identical on every numpy version, so golden fixtures
only have to store expected outputs, never the inputs.
"""
import numpy as np

from .config.coco_data import COCO_KEYPOINTS, COCO_PERSON_SKELETON, heatmap_hflip

_U64 = np.uint64
_MASK = (1 << 64) - 1


def _mix64(x):
    """splitmix64 finaliser on a uint64 array (wrap-around arithmetic)."""
    with np.errstate(over='ignore'):
        x = (x + _U64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> _U64(30))) * _U64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> _U64(27))) * _U64(0x94D049BB133111EB)
        return x ^ (x >> _U64(31))


class HashRng:
    """Counter-based generator: value i of stream s of seed k = f(k, s, i)."""

    def __init__(self, seed):
        self._key = _mix64(np.array([int(seed) & _MASK], dtype=_U64))[0]
        self._stream = 0

    def _bits(self, n):
        self._stream += 1
        with np.errstate(over='ignore'):
            base = _mix64(np.array([self._stream], dtype=_U64) ^ self._key)[0]
            return _mix64(np.arange(n, dtype=_U64) * _U64(0xD1342543DE82EF95) + base)

    def uniform(self, n, lo=0.0, hi=1.0):
        u = (self._bits(n) >> _U64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def integers(self, n, lo, hi):
        """Uniform integers in [lo, hi]."""
        return lo + (self._bits(n) % _U64(hi - lo + 1)).astype(np.int64)

    def normal(self, n):
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)  # (0, 1]
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)])
        return z[:n]


# unit-height stick figure: (x, y) with y down, x to the image's right, origin at the hip centre
_TEMPLATE = np.array([
    [0.00, -0.52],                   # nose
    [0.03, -0.55], [-0.03, -0.55],   # eyes  (left eye is on the image's right for a facing person)
    [0.06, -0.53], [-0.06, -0.53],   # ears
    [0.12, -0.40], [-0.12, -0.40],   # shoulders
    [0.16, -0.22], [-0.16, -0.22],   # elbows
    [0.18, -0.05], [-0.18, -0.05],   # wrists
    [0.08, 0.00], [-0.08, 0.00],     # hips
    [0.09, 0.22], [-0.09, 0.22],     # knees
    [0.10, 0.45], [-0.10, 0.45],     # ankles
], dtype=np.float64)


def make_scene(rng, height, width, n_persons=None):
    """Joint coordinates (P,17,2) in input pixels, visibility (P,17) and amplitudes (P,17)."""
    if n_persons is None:
        n_persons = int(rng.integers(1, 4, 20)[0])
    p = n_persons
    size = rng.uniform(p, 0.19, 0.59) * height
    cx = rng.uniform(p, 0.08, 0.92) * width
    cy = rng.uniform(p, 0.30, 0.70) * height
    lean = rng.uniform(p, -0.25, 0.25)
    jit = rng.normal(p * 17 * 2).reshape(p, 17, 2) * 3.0
    vis = rng.uniform(p * 17).reshape(p, 17) < 0.85
    amp = rng.uniform(p * 17, 0.3, 1.0).reshape(p, 17)
    t = _TEMPLATE[None] * size[:, None, None]
    x = cx[:, None] + t[..., 0] + lean[:, None] * t[..., 1]
    y = cy[:, None] + t[..., 1]
    xy = np.stack([x, y], -1) + jit
    inside = (xy[..., 0] > 2) & (xy[..., 0] < width - 3) & (xy[..., 1] > 2) & (xy[..., 1] < height - 3)
    return xy, vis & inside, amp


def mirror_scene(xy, vis, amp, width, keypoints=COCO_KEYPOINTS):
    """The same scene seen in the horizontally flipped image."""
    perm = heatmap_hflip(keypoints)
    xy_m = xy[:, perm].copy()
    xy_m[..., 0] = (width - 1) - xy_m[..., 0]
    return xy_m, vis[:, perm], amp[:, perm]


def render_maps(rng, xy, vis, amp, height, width, skeleton=COCO_PERSON_SKELETON, stride=4,
                sigma=7.0, patch=9, hm_noise=0.01, off_noise=0.5):
    """Low-res heatmaps (17,h,w) and offsets (2L,h,w), float32."""
    h, w = height // stride, width // stride
    gx = np.arange(w, dtype=np.float64) * stride + (stride / 2 - 0.5)
    gy = np.arange(h, dtype=np.float64) * stride + (stride / 2 - 0.5)
    n_kp = xy.shape[1]
    hm = np.zeros((n_kp, h, w), np.float64)
    for p in range(xy.shape[0]):
        for j in range(n_kp):
            if not vis[p, j]:
                continue
            ex = np.exp(-((gx - xy[p, j, 0]) ** 2) / (2 * sigma * sigma))
            ey = np.exp(-((gy - xy[p, j, 1]) ** 2) / (2 * sigma * sigma))
            np.maximum(hm[j], amp[p, j] * ey[:, None] * ex[None, :], out=hm[j])
    hm += rng.normal(hm.size).reshape(hm.shape) * hm_noise
    L = len(skeleton)
    off = np.zeros((2 * L, h, w), np.float64)
    r = patch // 2
    for p in range(xy.shape[0]):
        for l, (a, b) in enumerate(skeleton):
            if not (vis[p, a] and vis[p, b]):
                continue
            ci = int(np.clip(np.rint((xy[p, a, 0] - (stride / 2 - 0.5)) / stride), 0, w - 1))
            ri = int(np.clip(np.rint((xy[p, a, 1] - (stride / 2 - 0.5)) / stride), 0, h - 1))
            y0, y1, x0, x1 = max(ri - r, 0), min(ri + r + 1, h), max(ci - r, 0), min(ci + r + 1, w)
            off[2 * l, y0:y1, x0:x1] = xy[p, b, 0] - gx[None, x0:x1]
            off[2 * l + 1, y0:y1, x0:x1] = xy[p, b, 1] - gy[y0:y1, None]
    off += rng.normal(off.size).reshape(off.shape) * off_noise
    return hm.astype(np.float32), off.astype(np.float32)


def synth_batch(seed, batch, height=640, width=640, flip=False, n_persons=None,
                skeleton=COCO_PERSON_SKELETON, **kw):
    """Head outputs for `batch` images: hm (B,17,h,w), off (B,2L,h,w); B = 2*batch when flip.

    With flip=True the second half holds the maps of the mirrored scenes, i.e. what the
    network would output for ``torch.flip(images, [-1])`` (evaluate.py:211-212).
    """
    hms, offs, hms_f, offs_f = [], [], [], []
    for i in range(batch):
        rng = HashRng(seed * 1000003 + i)
        scene = make_scene(rng, height, width, n_persons)
        a, b = render_maps(rng, *scene, height, width, skeleton, **kw)
        hms.append(a)
        offs.append(b)
        if flip:
            a, b = render_maps(rng, *mirror_scene(*scene, width), height, width, skeleton, **kw)
            hms_f.append(a)
            offs_f.append(b)
    return np.stack(hms + hms_f), np.stack(offs + offs_f)


def noise_batch(seed, shape, scale=1.0):
    """Plain N(0, scale) tensor from the portable generator (adversarial / stress inputs)."""
    rng = HashRng(seed)
    n = int(np.prod(shape))
    return (rng.normal(n) * scale).reshape(shape).astype(np.float32)
