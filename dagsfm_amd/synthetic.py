"""Synthetic image sets of the shapes BASELINE.json names (no datasets are reachable offline).

A scene is a pool of 3-D points, each with a SIFT-like base descriptor made with the recipe
of the reference's matcher tests (/root/reference/src/feature/sift_test.cc:243-253: 128 x
U(0,1)^2, L2-normalised, x512, rounded, saturated to uint8).  Every image observes a random
subset of the pool through a random SIMPLE_PINHOLE camera (f=800, 1000x750, sigma=0.5 px
keypoint noise, small descriptor noise) and is filled up with clutter features (random
descriptors, uniform keypoints).  Two images share about n_obs^2 / n_pool true
correspondences (256 at the default 4096-feature shape), so matching has real matches to find
and two-view verification has real geometry to verify.
"""
import numpy as np


def _sift_like(rng, n):
    x = rng.random((n, 128), dtype=np.float32)
    x *= x
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x


def _to_u8(x):
    return np.clip(np.rint(512.0 * x), 0, 255).astype(np.uint8)


def _look_at(cam_pos, target, up):
    z = target - cam_pos
    z /= np.linalg.norm(z)
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=0)  # world -> camera
    t = -R @ cam_pos
    return R, t


def world_to_image(model_id, params, u, v):
    """<Model>::WorldToImage of /root/reference/src/base/camera_models.h for the eleven camera models (numpy, arrays):
    normalised camera coordinates -> pixels.  Only the test / bench scenes use it (to make keypoints that are
    consistent with a distorted camera); the product path only ever needs the inverse (ImageToWorld)."""
    p = [float(x) for x in params]
    two_focal = model_id in (1, 4, 5, 6, 7, 10)
    if two_focal:
        f1, f2, c1, c2, e = p[0], p[1], p[2], p[3], p[4:]
    else:
        f1, f2, c1, c2, e = p[0], p[0], p[1], p[2], p[3:]
    u = np.asarray(u, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    if model_id == 10:  # THIN_PRISM_FISHEYE lifts to the equidistant sphere first
        r = np.sqrt(u * u + v * v)
        sc = np.where(r > np.finfo(np.float64).eps, np.arctan(r) / np.where(r > 0, r, 1.0), 1.0)
        u, v = u * sc, v * sc
    u2, v2, uv = u * u, v * v, u * v
    r2 = u2 + v2
    if model_id in (0, 1):
        du, dv = 0.0 * u, 0.0 * v
    elif model_id == 2:
        du, dv = u * (e[0] * r2), v * (e[0] * r2)
    elif model_id == 3:
        rad = e[0] * r2 + e[1] * r2 * r2
        du, dv = u * rad, v * rad
    elif model_id == 4:
        rad = e[0] * r2 + e[1] * r2 * r2
        du = u * rad + 2 * e[2] * uv + e[3] * (r2 + 2 * u2)
        dv = v * rad + 2 * e[3] * uv + e[2] * (r2 + 2 * v2)
    elif model_id == 6:
        r4, r6 = r2 * r2, r2 * r2 * r2
        rad = (1 + e[0] * r2 + e[1] * r4 + e[4] * r6) / (1 + e[5] * r2 + e[6] * r4 + e[7] * r6)
        du = u * rad + 2 * e[2] * uv + e[3] * (r2 + 2 * u2) - u
        dv = v * rad + 2 * e[3] * uv + e[2] * (r2 + 2 * v2) - v
    elif model_id in (5, 8, 9):
        k = (list(e) + [0.0, 0.0, 0.0])[:4] if model_id != 5 else list(e[:4])
        r = np.sqrt(r2)
        th = np.arctan(r)
        th2 = th * th
        thd = th * (1 + k[0] * th2 + k[1] * th2 ** 2 + k[2] * th2 ** 3 + k[3] * th2 ** 4)
        ok = r > np.finfo(np.float64).eps
        rs = np.where(ok, r, 1.0)
        du = np.where(ok, u * thd / rs - u, 0.0)
        dv = np.where(ok, v * thd / rs - v, 0.0)
    elif model_id == 7:
        om = e[0]
        rad = np.sqrt(r2)
        if om * om < 1e-4:
            fac = (om * om * r2) / 3 - om * om / 12 + 1
        else:
            small = r2 < 1e-4
            th = np.tan(om / 2)
            f_small = (-2 * th * (4 * r2 * th * th - 3)) / (3 * om)
            f_big = np.arctan(np.where(small, 1.0, rad) * 2 * th) / (np.where(small, 1.0, rad) * om)
            fac = np.where(small, f_small, f_big)
        du, dv = u * fac - u, v * fac - v
    elif model_id == 10:
        r4, r6, r8 = r2 * r2, r2 ** 3, r2 ** 4
        rad = e[0] * r2 + e[1] * r4 + e[4] * r6 + e[5] * r8
        du = u * rad + 2 * e[2] * uv + e[3] * (r2 + 2 * u2) + e[6] * r2
        dv = v * rad + 2 * e[3] * uv + e[2] * (r2 + 2 * v2) + e[7] * r2
    else:
        raise ValueError("camera model %d does not exist" % model_id)
    return f1 * (u + du) + c1, f2 * (v + dv) + c2


class Scene:
    def __init__(self, n_images, n_feats, seed=0, n_obs=None, n_pool=None, focal=800.0, width=1000, height=750,
                 kp_sigma=0.5, desc_sigma=0.012, planar=False, outlier_frac=0.2, planar_depth=0.05, panoramic=False,
                 camera=None):
        """planar: the point pool is flattened to z in +-5 * planar_depth (0.0 = an exact plane); panoramic: every camera
        sits at the same centre (pure rotation between the images); camera = (model_id, params): keypoints are
        projected through that camera model's WorldToImage instead of the SIMPLE_PINHOLE default."""
        self.panoramic = panoramic
        self.camera = camera
        self.n_images, self.n_feats, self.seed = n_images, n_feats, seed
        self.n_obs = n_obs if n_obs is not None else n_feats // 2
        self.n_pool = n_pool if n_pool is not None else max(4 * n_feats, self.n_obs)
        self.focal, self.width, self.height = focal, width, height
        self.kp_sigma, self.desc_sigma = kp_sigma, desc_sigma
        self.outlier_frac = outlier_frac
        rng = np.random.default_rng([seed, 0xD5F])
        self.points = rng.uniform(-5.0, 5.0, (self.n_pool, 3))
        if planar:
            self.points[:, 2] = planar_depth * self.points[:, 2]
        self.base = _sift_like(rng, self.n_pool)

    def _pose(self, rng):
        # camera on a shell around the scene, looking at a jittered centre
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        d[2] = -abs(d[2]) - 0.6  # keep cameras on one side so that planar scenes stay visible
        d /= np.linalg.norm(d)
        pos = d * rng.uniform(14.0, 20.0)
        target = rng.uniform(-1.0, 1.0, 3)
        if self.panoramic:  # one shared centre, the viewing direction is what differs
            pos = np.array([0.0, 0.0, -17.0])
            target = 3.0 * target
        return _look_at(pos, target, np.array([0.0, 1.0, 0.0]) + 0.2 * rng.normal(size=3))

    def pose(self, i):
        """(R, t) world -> camera of image i without generating its features."""
        return self._pose(np.random.default_rng([self.seed, 1 + i]))

    def image(self, i):
        """Returns (descriptors u8 [n,128], keypoints f32 [n,2], (R,t), observed pool ids [n] (-1 = clutter))."""
        rng = np.random.default_rng([self.seed, 1 + i])
        n, k = self.n_feats, min(self.n_obs, self.n_feats)
        R, t = self._pose(rng)
        ids = np.full(n, -1, dtype=np.int64)
        obs = rng.choice(self.n_pool, size=k, replace=False)
        desc = np.empty((n, 128), dtype=np.float32)
        kp = np.empty((n, 2), dtype=np.float64)
        pc = self.points[obs] @ R.T + t
        if self.camera is None:
            kp[:k, 0] = self.focal * pc[:, 0] / pc[:, 2] + self.width / 2.0
            kp[:k, 1] = self.focal * pc[:, 1] / pc[:, 2] + self.height / 2.0
        else:
            kp[:k, 0], kp[:k, 1] = world_to_image(self.camera[0], self.camera[1], pc[:, 0] / pc[:, 2], pc[:, 1] / pc[:, 2])
        kp[:k] += rng.normal(scale=self.kp_sigma, size=(k, 2))
        # geometric outliers: the descriptor still matches but the keypoint is somewhere else
        bad = rng.random(k) < self.outlier_frac
        kp[:k][bad, 0] = rng.uniform(0.0, self.width, int(bad.sum()))
        kp[:k][bad, 1] = rng.uniform(0.0, self.height, int(bad.sum()))
        d_obs = self.base[obs] + rng.normal(scale=self.desc_sigma, size=(k, 128)).astype(np.float32)
        np.maximum(d_obs, 0.0, out=d_obs)
        d_obs /= np.linalg.norm(d_obs, axis=1, keepdims=True)
        desc[:k] = d_obs
        ids[:k] = obs
        if n > k:
            desc[k:] = _sift_like(rng, n - k)
            kp[k:, 0] = rng.uniform(0.0, self.width, n - k)
            kp[k:, 1] = rng.uniform(0.0, self.height, n - k)
        perm = rng.permutation(n)
        return _to_u8(desc[perm]), kp[perm].astype(np.float32), (R, t), ids[perm]

    def camera_params(self):
        """SIMPLE_PINHOLE (model_id 0): f, cx, cy."""
        return [self.focal, self.width / 2.0, self.height / 2.0]


def exhaustive_pairs(n_images):
    """All image pairs i<j in the order ExhaustiveFeatureMatcher visits a single block
    (/root/reference/src/feature/matching.cc:853-915 with one block: idx1 < idx2)."""
    i, j = np.triu_indices(n_images, k=1)
    return np.stack([i, j], axis=1).astype(np.uint32)


def vocabulary(scene, num_words, seed=0):
    """A vocabulary for the retrieval tests / bench (no pre-trained vocab tree is reachable offline): visual words =
    `num_words` descriptors of the scene's pool with a little noise (what hierarchical k-means centres look like for
    this generator), an orthonormal 64 x 128 projection from the QR of a seeded Gaussian matrix
    (InvertedIndex::GenerateHammingEmbeddingProjection, /root/reference/src/retrieval/inverted_index.h:175-184) and
    per-word thresholds = the projection of the word itself plus noise (stand-in for the per-word medians of
    ComputeHammingEmbedding, :186-217).  Returns (words u8 [W,128], projection f32 [64,128], thresholds f32 [W,64])."""
    rng = np.random.default_rng([scene.seed, 0x70CAB, seed])
    pick = rng.choice(scene.n_pool, size=num_words, replace=num_words > scene.n_pool)
    w = scene.base[pick] + rng.normal(scale=0.02, size=(num_words, 128)).astype(np.float32)
    np.maximum(w, 0.0, out=w)
    w /= np.linalg.norm(w, axis=1, keepdims=True)
    words = _to_u8(w)
    q, _ = np.linalg.qr(rng.normal(size=(128, 128)))
    proj = np.ascontiguousarray(q[:64].astype(np.float32))
    thr = (words.astype(np.float32) @ proj.T + rng.normal(scale=8.0, size=(num_words, 64))).astype(np.float32)
    return words, proj, np.ascontiguousarray(thr)


def knn_pairs(scene, n_images, k, seed):
    """Candidate-pair graph standing in for the vocabulary-tree retrieval of configs[3]: the K
    nearest images of every image by camera-centre distance (seeded scene => deterministic), deduplicated to id1 < id2
    and sorted -- the order VocabTreeFeatureMatcher hands its pairs to Match() in (by query image)."""
    centres = np.empty((n_images, 3))
    for i in range(n_images):
        R, t = scene.pose(i)
        centres[i] = -R.T @ t
    k = min(k, n_images - 1)
    pairs = set()
    blk = 1024
    for s in range(0, n_images, blk):
        d = ((centres[s:s + blk, None, :] - centres[None, :, :]) ** 2).sum(-1)
        d[np.arange(d.shape[0]), np.arange(s, s + d.shape[0])] = np.inf
        nn = np.argpartition(d, k - 1, axis=1)[:, :k]
        for r in range(nn.shape[0]):
            i = s + r
            for j in nn[r]:
                pairs.add((min(i, int(j)), max(i, int(j))))
    out = np.array(sorted(pairs), dtype=np.uint32).reshape(-1, 2)
    return out
