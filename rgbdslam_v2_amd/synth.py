"""Seeded synthetic RGB-D front-end data ("Level A").

A world of 3-D points with random 256-bit descriptors is observed from a smooth
camera trajectory.  Every frame holds exactly ``n_kp`` keypoints: observed world
points (descriptor bits flipped independently per observation, pixel + depth
noise, back-projected exactly like Node::projectTo3D does, node.cpp:900-965)
padded with outliers (random descriptors at random places).  Used by the tests
(small sizes) and by bench.py (BASELINE.json configs[1]: 640x480, ORB 1000 kp,
20 candidate pairs per frame).
"""
import numpy as np

FX = FY = 525.0
CX, CY = 319.5, 239.5
WIDTH, HEIGHT = 640, 480
DEPTH_NOISE = 0.01      # sigma = sigma_depth * z^2
DEPTH_NOISE_R1 = 0.002   # round 1's regime, kept as a second test / bench parametrisation


def _rot(rx, ry, rz):
    cx, sx = np.cos(rx), np.sin(rx)
    cy, sy = np.cos(ry), np.sin(ry)
    cz, sz = np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_sequence(n_frames=200, n_kp=1000, n_world=4000, seed=20260923, bitflip=0.08,
                  true_fraction=0.75, depth_noise=0.01, pixel_noise=0.3, nan_fraction=0.0,
                  width=WIDTH, height=HEIGHT, motion_scale=1.0):
    """Returns dict(desc uint8 [F,N,32], xyz1 float32 [F,N,4], poses float64 [F,4,4]
    (camera-to-world), world_id int32 [F,N] (-1 = outlier)).

    depth_noise is the sigma of the depth error as a multiple of z^2: 0.01 = sigma_depth
    (parameter_server.cpp:46), what SURVEY.md 8(d) specifies and bench.py measures; DEPTH_NOISE_R1
    = 0.002 is the gentler regime round 1 measured (more valid hypotheses per pair, a slower
    RANSAC stage) that the tests keep as a second parametrisation."""
    rng = np.random.Generator(np.random.PCG64(seed))
    fx, fy = FX * width / WIDTH, FY * height / HEIGHT
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    world = np.empty((n_world, 3))
    world[:, 0] = rng.uniform(-2.0, 2.0, n_world)
    world[:, 1] = rng.uniform(-1.5, 1.5, n_world)
    world[:, 2] = rng.uniform(0.8, 4.0, n_world)
    wdesc = rng.integers(0, 256, size=(n_world, 32), dtype=np.uint8)
    # detector repeatability: every world point has a fixed "response"; a frame keeps the
    # strongest visible ones (cv::KeyPointsFilter::retainBest, node.cpp:187-191), with a
    # little per-frame jitter.
    response = rng.random(n_world)

    # smooth bounded trajectory: <= ~5 cm and <= ~3 deg between consecutive frames
    f = np.arange(n_frames)
    ph = rng.uniform(0, 2 * np.pi, 6)
    w = 2 * np.pi / 140.0
    tx = 0.55 * motion_scale * np.sin(w * f + ph[0])
    ty = 0.30 * motion_scale * np.sin(0.7 * w * f + ph[1])
    tz = 0.25 * motion_scale * np.sin(1.3 * w * f + ph[2])
    rx = np.deg2rad(6.0) * motion_scale * np.sin(0.9 * w * f + ph[3])
    ry = np.deg2rad(14.0) * motion_scale * np.sin(1.1 * w * f + ph[4])
    rz = np.deg2rad(5.0) * motion_scale * np.sin(0.6 * w * f + ph[5])

    desc = np.empty((n_frames, n_kp, 32), np.uint8)
    xyz1 = np.empty((n_frames, n_kp, 4), np.float32)
    poses = np.empty((n_frames, 4, 4))
    world_id = np.full((n_frames, n_kp), -1, np.int32)
    n_true_max = int(round(true_fraction * n_kp))
    for k in range(n_frames):
        R = _rot(rx[k], ry[k], rz[k])
        t = np.array([tx[k], ty[k], tz[k]])
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = t
        poses[k] = T
        pc = (world - t) @ R  # R^T (p - t)
        z = pc[:, 2]
        u = fx * pc[:, 0] / np.maximum(z, 1e-6) + cx
        v = fy * pc[:, 1] / np.maximum(z, 1e-6) + cy
        vis = np.flatnonzero((z > 0.5) & (z < 5.0) & (u >= 1) & (u < width - 1) &
                             (v >= 1) & (v < height - 1))
        n_true = min(n_true_max, vis.size)
        score = response[vis] + rng.normal(0, 0.08, vis.size)
        pick = vis[np.argsort(-score, kind="stable")[:n_true]]
        # observation model: pixel noise, depth noise, float32 back-projection
        uu = (u[pick] + rng.normal(0, pixel_noise, n_true)).astype(np.float32)
        vv = (v[pick] + rng.normal(0, pixel_noise, n_true)).astype(np.float32)
        zz = (z[pick] + rng.normal(0, 1, n_true) * depth_noise * z[pick] ** 2).astype(np.float32)
        flips = rng.random((n_true, 256)) < bitflip
        d_true = wdesc[pick] ^ np.packbits(flips, axis=1, bitorder="little")
        # outliers
        n_out = n_kp - n_true
        uo = rng.uniform(1, width - 1, n_out).astype(np.float32)
        vo = rng.uniform(1, height - 1, n_out).astype(np.float32)
        zo = rng.uniform(0.6, 4.5, n_out).astype(np.float32)
        d_out = rng.integers(0, 256, size=(n_out, 32), dtype=np.uint8)
        U = np.concatenate([uu, uo])
        V = np.concatenate([vv, vo])
        Z = np.concatenate([zz, zo])
        D = np.concatenate([d_true, d_out])
        wid = np.concatenate([pick.astype(np.int32), np.full(n_out, -1, np.int32)])
        if nan_fraction > 0:
            Z = Z.copy()
            Z[rng.random(n_kp) < nan_fraction] = np.nan
        perm = rng.permutation(n_kp)
        U, V, Z, D, wid = U[perm], V[perm], Z[perm], D[perm], wid[perm]
        fxinv = np.float32(1.0 / fx)
        fyinv = np.float32(1.0 / fy)
        X = (U - np.float32(cx)) * Z * fxinv  # misc2.h:62
        Y = (V - np.float32(cy)) * Z * fyinv  # misc2.h:63
        xyz1[k, :, 0] = X
        xyz1[k, :, 1] = Y
        xyz1[k, :, 2] = Z
        xyz1[k, :, 3] = 1.0
        desc[k] = D
        world_id[k] = wid
    return dict(desc=desc, xyz1=xyz1, poses=poses, world_id=world_id)


def candidate_pairs(n_frames, per_frame=20, seed=20260923, predecessors=3):
    """For every frame: `predecessors` sequential predecessors (wrapping) plus pseudo-random
    other frames up to `per_frame` candidates (200 x 20 = 4000 pairs)."""
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    per_frame = min(per_frame, n_frames - 1)
    pq, pt = [], []
    for f in range(n_frames):
        cands = [(f - d) % n_frames for d in range(1, min(predecessors, per_frame) + 1)]
        pool = np.array([c for c in range(n_frames) if c != f and c not in cands])
        extra = rng.choice(pool, size=per_frame - len(cands), replace=False) if per_frame > len(cands) else []
        for c in list(cands) + [int(e) for e in extra]:
            pq.append(f)
            pt.append(c)
    return np.asarray(pq, np.int32), np.asarray(pt, np.int32)


def loop_closure_places(n_frames=180, n_kp=1000, frames_per_place=10, depth_noise=DEPTH_NOISE, seed=1000):
    """The reject-path workload of bench.py's `loop_closure` sub-record: `n_frames` frames in unrelated places of
    `frames_per_place` frames each, every frame against every earlier one (180 frames -> 16 110 pairs, ~5 % of them
    true edges; the others are chance matches that the min_matches gate or RANSAC rejects).
    Returns (desc list, xyz1 list, pair_q, pair_t)."""
    places = [make_sequence(n_frames=frames_per_place, n_kp=n_kp, seed=seed + p, depth_noise=depth_noise)
              for p in range(n_frames // frames_per_place)]
    desc = [pl["desc"][i] for pl in places for i in range(frames_per_place)]
    xyz = [pl["xyz1"][i] for pl in places for i in range(frames_per_place)]
    F = len(desc)
    pq = np.array([q for q in range(F) for t in range(q)], np.int32)
    pt = np.array([t for q in range(F) for t in range(q)], np.int32)
    return desc, xyz, pq, pt


def relative_pose(poses, q, t):
    """Ground-truth transform mapping frame q's camera coordinates into frame t's."""
    return np.linalg.inv(poses[t]) @ poses[q]


def sift_descriptors_like(desc_bits: np.ndarray, seed: int = 0, noise: float = 0.04) -> np.ndarray:
    """128-d float descriptors for BASELINE configs[3] derived from the binary ones of make_sequence:
    every 256-bit descriptor is mapped to a fixed pseudo-random non-negative 128-vector (pairs of
    bits -> magnitude), perturbed, clamped at 0.2 and L2-normalised like SIFT; observations of the
    same world point (few flipped bits) stay close, unrelated ones are far apart."""
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    bits = np.unpackbits(desc_bits, axis=-1, bitorder="little").astype(np.float32)  # [..., 256]
    v = bits[..., 0::2] * 2.0 + bits[..., 1::2] + 0.25                               # [..., 128]
    v = v + rng.normal(0, noise, v.shape).astype(np.float32)
    v = np.maximum(v, 0)
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    v = np.minimum(v, 0.2)
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    return v.astype(np.float32)


def forth_and_back(n_frames, base_len):
    """Indices into a generated sequence of base_len frames that walk it forth and back: a continuous camera path of any
    length (long batches for rgbdfe_detect_describe_batch without generating every frame)."""
    return [i % base_len if (i // base_len) % 2 == 0 else base_len - 1 - i % base_len for i in range(n_frames)]


def make_image_sequence(n_frames=4, width=WIDTH, height=HEIGHT, seed=20260923, plane_depth=2.0,
                        nan_fraction=0.03):
    """SURVEY.md 8(d) "Level B": textured gray images of a fronto-parallel plane seen from a moving
    camera (pure translation + in-plane rotation -> an exact similarity warp of a large random
    blob texture), float depth = plane depth (with NaN holes), mono8 mask as the reference derives
    it (depthToCV8UC1, misc.cpp:414-430: convertTo(CV_8UC1, 100) -> 0 where depth is NaN)."""
    rng = np.random.Generator(np.random.PCG64(seed + 5))
    T = 2048
    tex = np.zeros((T, T), np.float32)
    # random blobs at several scales -> corners at several pyramid levels
    for scale, count in ((3, 9000), (6, 3000), (12, 900), (25, 250)):
        cx = rng.integers(0, T, count); cy = rng.integers(0, T, count)
        amp = rng.uniform(-90, 90, count)
        for x, y, a in zip(cx, cy, amp):
            x0, x1 = max(x - scale, 0), min(x + scale, T)
            y0, y1 = max(y - scale, 0), min(y + scale, T)
            tex[y0:y1, x0:x1] += a
    tex = np.clip(tex + 128, 0, 255)
    imgs, depths, masks = [], [], []
    f = FX * width / WIDTH
    for k in range(n_frames):
        ang = np.deg2rad(4.0) * np.sin(0.7 * k + 0.3)
        s = 1.0 + 0.04 * np.sin(0.5 * k)
        tx, ty = 600 + 25 * k, 700 + 12 * k
        v, u = np.mgrid[0:height, 0:width].astype(np.float32)
        uc, vc = u - (width - 1) / 2, v - (height - 1) / 2
        xs = s * (np.cos(ang) * uc - np.sin(ang) * vc) + tx
        ys = s * (np.sin(ang) * uc + np.cos(ang) * vc) + ty
        x0 = np.clip(np.floor(xs).astype(np.int64), 0, T - 2); y0 = np.clip(np.floor(ys).astype(np.int64), 0, T - 2)
        fx_, fy_ = xs - x0, ys - y0
        img = (tex[y0, x0] * (1 - fx_) * (1 - fy_) + tex[y0, x0 + 1] * fx_ * (1 - fy_) +
               tex[y0 + 1, x0] * (1 - fx_) * fy_ + tex[y0 + 1, x0 + 1] * fx_ * fy_)
        img = img + rng.normal(0, 2.0, img.shape)
        imgs.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
        d = np.full((height, width), plane_depth * s, np.float32)
        holes = rng.random((height // 16 + 1, width // 16 + 1)) < nan_fraction
        d[np.kron(holes, np.ones((16, 16), bool))[:height, :width]] = np.nan
        depths.append(d)
        m = np.where(np.isnan(d), 0, np.clip(np.rint(np.nan_to_num(d) * 100), 0, 255)).astype(np.uint8)
        masks.append(m)
    return dict(gray=np.stack(imgs), depth=np.stack(depths), mask=np.stack(masks), fx=f, fy=f,
                cx=(width - 1) / 2.0, cy=(height - 1) / 2.0)


def make_depth_sequence(n_frames=4, width=WIDTH, height=HEIGHT, seed=20260923, nan_fraction=0.03, noise=0.002):
    """Depth images of one static scene (a wavy wall at ~2 m with a box in front of it) seen from a
    translating camera: geometrically consistent frames for the environment measurement model
    (observationLikelihood, misc.cpp:814-969).  Returns depth [F, H, W] f32 (NaN holes), poses [F, 4, 4]
    (camera-to-world), intrinsics.  relative_pose(poses, new, old) is the new -> old transform."""
    rng = np.random.Generator(np.random.PCG64(seed + 9))
    f = FX * width / WIDTH
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    v, u = np.mgrid[0:height, 0:width].astype(np.float64)
    rx, ry = (u - cx) / f, (v - cy) / f

    def wall(x, y):
        return 2.0 + 0.25 * np.sin(1.7 * x) * np.cos(1.3 * y)

    depths, poses = [], []
    for k in range(n_frames):
        t = np.array([0.04 * k, 0.02 * np.sin(0.9 * k), 0.03 * np.cos(0.7 * k) - 0.03])
        z = np.full((height, width), 2.0)
        for _ in range(12):  # fixed point of z = wall(x_w, y_w) - t_z along each pixel ray
            z = wall(rx * z + t[0], ry * z + t[1]) - t[2]
        # a box 0.5 m in front of the wall, fixed in the world: |x_w - 0.1| < 0.25, |y_w + 0.05| < 0.2 at z_w = 1.4
        zb = 1.4 - t[2]
        inside = (np.abs(rx * zb + t[0] - 0.1) < 0.25) & (np.abs(ry * zb + t[1] + 0.05) < 0.2)
        z = np.where(inside, zb, z)
        z = z + rng.normal(0, 1.0, z.shape) * noise * z * z
        d = z.astype(np.float32)
        holes = rng.random((height // 16 + 1, width // 16 + 1)) < nan_fraction
        d[np.kron(holes, np.ones((16, 16), bool))[:height, :width]] = np.nan
        depths.append(d)
        P = np.eye(4)
        P[:3, 3] = t
        poses.append(P)
    return dict(depth=np.stack(depths), poses=np.stack(poses), fx=f, fy=f, cx=cx, cy=cy)
