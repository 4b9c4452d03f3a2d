"""Seeded synthetic maps and queries in the reference's data model.

A map is what ExtFeatAndMatch + OpenMVG leave on disk for OpenMVGLocalization_AKAZE: per view a list of
64-byte M-LDB rows (.desc) with keypoints (.feat), a structure of landmarks observed by (view, feat)
pairs (sfm_data.json), one pinhole intrinsic.  Geometry is consistent (keypoints are projections of the
landmarks), so every stage of the path has a planted answer: the query's true pose, the landmarks it
sees and which of its descriptor copies sit at wrong image positions (outliers).

Descriptor bytes: uniform random with bytes 61..63 and the top two bits of byte 60 zero (486 valid bits,
FileUtils.cpp:77-92).  Test and benchmark infrastructure: it lives outside the product package (sfmlocalization_amd/), which never imports it.
"""
from dataclasses import dataclass, field

import numpy as np

VALID_BITS = 486


def random_descriptors(rng, n):
    d = rng.integers(0, 256, size=(n, 64), dtype=np.uint8)
    d[:, 61:] = 0
    d[:, 60] &= 0x3F
    return d


def flip_bits(rng, desc, max_flips):
    """XOR up to max_flips random valid bit positions into each row (duplicates cancel)."""
    n = desc.shape[0]
    if n == 0 or max_flips <= 0:
        return desc.copy()
    k = rng.integers(0, max_flips + 1, size=n)
    pos = rng.integers(0, VALID_BITS, size=(n, max_flips))
    use = np.arange(max_flips)[None, :] < k[:, None]
    rows = np.broadcast_to(np.arange(n)[:, None], pos.shape)[use]
    p = pos[use]
    out = desc.copy()
    np.bitwise_xor.at(out, (rows, p >> 3), (1 << (p & 7)).astype(np.uint8))
    return out


def look_at(C, target, rng=None, jitter=0.0):
    """World->camera rotations (rows = camera axes) for centres C [n,3] looking at target [n,3]."""
    z = target - C
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    if rng is not None and jitter > 0:
        z = z + rng.normal(0, jitter, size=z.shape)
        z /= np.linalg.norm(z, axis=1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    x = np.cross(np.broadcast_to(up, z.shape), z)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = np.cross(z, x)
    return np.stack([x, y, z], axis=1)


def project(R, C, X, f, ppx, ppy):
    """Pinhole projection x = K (R (X - C)); returns pixel [n,2] and depth [n]."""
    Xc = (X - C) @ R.T
    z = Xc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = f * Xc[:, 0] / z + ppx
        v = f * Xc[:, 1] / z + ppy
    return np.stack([u, v], axis=1), z


def round6(a):
    """What a float keeps after `ostream << float` at default precision and a read back
    (AKAZEOpenCV.cpp:80-81 writes .feat that way; Regions::Load reads it back, :106-111)."""
    a = np.asarray(a, dtype=np.float32)
    flat = np.array([np.float32(float("%.6g" % v)) for v in a.ravel()], dtype=np.float32)
    return flat.reshape(a.shape)


@dataclass
class SynthMap:
    view_id: np.ndarray
    view_off: np.ndarray
    view_wh: np.ndarray
    desc: np.ndarray            # [n_rows, 64] u8
    kpt_xy: np.ndarray          # [n_rows, 2] f32
    row_landmark: np.ndarray    # [n_rows] i32 slot or -1
    landmark_id: np.ndarray     # [L] u32
    landmark_X: np.ndarray      # [L, 3] f64
    landmark_desc: np.ndarray   # [L, 64] u8 (generator's base descriptor, not part of the contract)
    landmark_place: np.ndarray  # [L]
    view_place: np.ndarray      # [V]
    view_R: np.ndarray          # [V,3,3]
    view_C: np.ndarray          # [V,3]
    place_center: np.ndarray    # [P,3]
    intrinsic: tuple            # (focal, ppx, ppy)
    width: int = 640
    height: int = 480
    extra: dict = field(default_factory=dict)

    @property
    def n_views(self):
        return len(self.view_id)

    @property
    def n_rows(self):
        return self.desc.shape[0]


def make_map(seed, n_views, desc_per_view=2000, views_per_place=20, landmarks_per_place=600,
             obs_per_view=250, map_flips=12, width=640, height=480, focal=800.0, kpt_noise=0.25,
             ragged=False, view_id_stride=1):
    """V views of `desc_per_view` rows; each view observes up to `obs_per_view` landmarks of its place,
    the rest of its rows are clutter (random descriptor, random keypoint, no landmark)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ppx, ppy = width / 2.0, height / 2.0
    n_places = max(1, (n_views + views_per_place - 1) // views_per_place)
    side = int(np.ceil(np.sqrt(n_places)))
    pc = np.array([[30.0 * (p % side), 30.0 * (p // side), 0.0] for p in range(n_places)])
    L = n_places * landmarks_per_place
    lm_place = np.repeat(np.arange(n_places), landmarks_per_place)
    lm_X = pc[lm_place] + rng.uniform(-5.0, 5.0, size=(L, 3))
    lm_desc = random_descriptors(rng, L)
    lm_id = (np.arange(L, dtype=np.uint32) * 3 + 7).astype(np.uint32)  # structure keys need not be dense

    view_place = (np.arange(n_views) // views_per_place).astype(np.int64)
    # camera centres on a ring/cap 12..16 m from the place centre
    az = rng.uniform(0, 2 * np.pi, n_views)
    el = rng.uniform(-0.3, 0.6, n_views)
    dist = rng.uniform(12.0, 16.0, n_views)
    dirs = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
    vC = pc[view_place] + dirs * dist[:, None]
    vR = look_at(vC, pc[view_place].copy(), rng, jitter=0.03)

    if ragged:
        counts = rng.integers(max(1, desc_per_view // 2), desc_per_view + 1, size=n_views)
        counts[rng.integers(0, n_views)] = 0 if n_views > 3 else counts[0]
    else:
        counts = np.full(n_views, desc_per_view)
    view_off = np.zeros(n_views + 1, dtype=np.uint32)
    view_off[1:] = np.cumsum(counts)
    n_rows = int(view_off[-1])

    desc = random_descriptors(rng, n_rows)
    kpt = np.stack([rng.uniform(0, width, n_rows), rng.uniform(0, height, n_rows)], axis=1).astype(np.float32)
    row_lm = np.full(n_rows, -1, dtype=np.int32)

    for v in range(n_views):
        n_v = int(counts[v])
        if n_v == 0:
            continue
        p = int(view_place[v])
        cand = np.arange(p * landmarks_per_place, (p + 1) * landmarks_per_place)
        px, z = project(vR[v], vC[v], lm_X[cand], focal, ppx, ppy)
        ok = (z > 0.5) & (px[:, 0] >= 1) & (px[:, 0] < width - 1) & (px[:, 1] >= 1) & (px[:, 1] < height - 1)
        vis = cand[ok]
        pxv = px[ok]
        n_obs = min(len(vis), obs_per_view, n_v)
        if n_obs == 0:
            continue
        pick = rng.choice(len(vis), size=n_obs, replace=False)
        rows = int(view_off[v]) + rng.choice(n_v, size=n_obs, replace=False)
        row_lm[rows] = vis[pick]
        kpt[rows] = (pxv[pick] + rng.normal(0, kpt_noise, size=(n_obs, 2))).astype(np.float32)
        desc[rows] = flip_bits(rng, lm_desc[vis[pick]], map_flips)
    kpt = round6(kpt) if n_rows <= 200000 else kpt  # .feat text round trip (exact for small maps; skipped for bench-size maps)
    view_id = (np.arange(n_views, dtype=np.uint32) * view_id_stride).astype(np.uint32)
    view_wh = np.tile(np.array([[width, height]], dtype=np.uint32), (n_views, 1))
    return SynthMap(view_id=view_id, view_off=view_off, view_wh=view_wh, desc=desc, kpt_xy=kpt,
                    row_landmark=row_lm, landmark_id=lm_id, landmark_X=lm_X, landmark_desc=lm_desc,
                    landmark_place=lm_place, view_place=view_place, view_R=vR, view_C=vC, place_center=pc,
                    intrinsic=(focal, ppx, ppy), width=width, height=height)


@dataclass
class SynthQuery:
    desc: np.ndarray          # [n, 64]
    kpt_xy: np.ndarray        # [n, 2] f32 (full precision, as locFeat, AKAZEOpenCV.cpp:77-79)
    width: int
    height: int
    R_true: np.ndarray
    C_true: np.ndarray
    place: int
    landmark: np.ndarray      # [n] landmark slot the descriptor was copied from, or -1
    is_inlier: np.ndarray     # [n] bool: copy sits at its true projection


def make_query(m: SynthMap, seed, n_feat=2000, n_copies=300, outlier_frac=0.3, query_flips=40,
               noise_px=1.0, place=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    f, ppx, ppy = m.intrinsic[:3]
    n_places = len(m.place_center)
    p = int(rng.integers(0, n_places)) if place is None else int(place)
    az, el, dist = rng.uniform(0, 2 * np.pi), rng.uniform(-0.2, 0.5), rng.uniform(12.0, 16.0)
    d = np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    C = m.place_center[p] + d * dist
    R = look_at(C[None, :], m.place_center[p][None, :].copy(), rng, jitter=0.03)[0]
    cand = np.nonzero(m.landmark_place == p)[0]
    px, z = project(R, C, m.landmark_X[cand], f, ppx, ppy)
    ok = (z > 0.5) & (px[:, 0] >= 1) & (px[:, 0] < m.width - 1) & (px[:, 1] >= 1) & (px[:, 1] < m.height - 1)
    vis, pxv = cand[ok], px[ok]
    n_copies = min(n_copies, len(vis), n_feat)
    pick = rng.choice(len(vis), size=n_copies, replace=False) if n_copies else np.zeros(0, np.int64)
    n_out = int(round(n_copies * outlier_frac))
    desc = random_descriptors(rng, n_feat)
    kpt = np.stack([rng.uniform(0, m.width, n_feat), rng.uniform(0, m.height, n_feat)], axis=1)
    lm = np.full(n_feat, -1, dtype=np.int64)
    inl = np.zeros(n_feat, dtype=bool)
    rows = rng.choice(n_feat, size=n_copies, replace=False) if n_copies else np.zeros(0, np.int64)
    desc[rows] = flip_bits(rng, m.landmark_desc[vis[pick]], query_flips)
    lm[rows] = vis[pick]
    true_px = pxv[pick] + rng.normal(0, noise_px, size=(n_copies, 2))
    is_out = np.zeros(n_copies, dtype=bool)
    is_out[:n_out] = True
    rng.shuffle(is_out)
    kpt[rows[~is_out]] = true_px[~is_out]
    inl[rows[~is_out]] = True
    return SynthQuery(desc=desc, kpt_xy=kpt.astype(np.float32), width=m.width, height=m.height, R_true=R,
                      C_true=C, place=p, landmark=lm, is_inlier=inl)


def planted_bank(seed, n_rows=512, n_query=96):
    """Small bank with planted near-duplicates at known distances, exact ties and d1 = 0 cases
    (SURVEY.md 8c golden (i)).  Returns (query, bank)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    q = random_descriptors(rng, n_query)
    bank = random_descriptors(rng, n_rows)

    def flipped(row, k):
        out = row.copy()
        pos = rng.choice(VALID_BITS, size=k, replace=False)
        for p_ in pos:
            out[p_ >> 3] ^= np.uint8(1 << (p_ & 7))
        return out

    # query-side plants first (so that bank rows derived below stay consistent)
    q[10] = q[3]           # exact ties: three identical query rows
    q[50] = q[3]
    a = q[20].copy()
    b = flipped(a, 20)
    q[21] = b              # rows 20/21 differ in exactly 20 bits

    r = 0
    for k in (0, 1, 5, 17, 40, 80, 120, 150):  # near-duplicates of query rows at known distances
        for _ in range(8):
            bank[r] = flipped(q[rng.integers(0, n_query)], k)
            r += 1
    # d0 == d1 == k, nearest must be the lower index (3, then 10)
    for k in (0, 4, 33):
        bank[r] = flipped(q[3], k)
        r += 1
    # equidistant from two different rows: 10 bits from a (index 20) and 10 from b (index 21)
    mid = a.copy()
    diff = np.unpackbits(a ^ b, bitorder="little")
    idx = np.nonzero(diff)[0][:10]
    for p_ in idx:
        mid[p_ >> 3] ^= np.uint8(1 << (p_ & 7))
    bank[r] = mid
    r += 1
    return q, bank


def write_map_to_disk(m: SynthMap, sfm_dir, match_dir, unposed_views=(), with_bow=None):
    """Lay the synthetic map out as the reference's tools would:
    <sfm_dir>/sfm_data.json, <match_dir>/image_describer.txt, <match_dir>/<base>.{desc,feat[,bow]}.
    `unposed_views`: view ids written without an extrinsic (they must be ignored by the localiser).
    Returns the list of image basenames."""
    import os
    from sfmlocalization_amd import fileio
    os.makedirs(sfm_dir, exist_ok=True)
    os.makedirs(match_dir, exist_ok=True)
    names = [f"img{int(v):06d}" for v in m.view_id]
    f, ppx, ppy = m.intrinsic[:3]
    poses = {int(v): (m.view_R[k], m.view_C[k]) for k, v in enumerate(m.view_id) if int(v) not in set(unposed_views)}
    slot_view = np.searchsorted(m.view_off, np.arange(m.n_rows), side="right") - 1
    obs_rows = np.nonzero(m.row_landmark >= 0)[0]
    per_lm = {}
    for r in obs_rows:
        k = int(slot_view[r])
        per_lm.setdefault(int(m.row_landmark[r]), []).append(
            (int(m.view_id[k]), int(r - m.view_off[k]), (float(m.kpt_xy[r, 0]), float(m.kpt_xy[r, 1]))))
    structure = [(int(m.landmark_id[s]), m.landmark_X[s], obs) for s, obs in sorted(per_lm.items())]
    # landmarks nobody observes still exist in the structure
    seen = set(per_lm)
    structure += [(int(m.landmark_id[s]), m.landmark_X[s], []) for s in range(len(m.landmark_id)) if s not in seen]
    structure.sort(key=lambda t: t[0])
    radial = len(m.intrinsic) >= 6   # (f, ppx, ppy, k1, k2, k3) -> a pinhole_radial_k3 intrinsic
    sd = fileio.make_sfm_data([int(v) for v in m.view_id], [n + ".jpg" for n in names], m.width, m.height, f, ppx,
                              ppy, poses=poses, structure=structure, root_path=os.path.abspath(sfm_dir),
                              intrinsic_type="pinhole_radial_k3" if radial else "pinhole",
                              disto_k3=tuple(m.intrinsic[3:6]) if radial else None)
    fileio.write_sfm_data(os.path.join(sfm_dir, "sfm_data.json"), sd)
    fileio.write_image_describer(os.path.join(match_dir, "image_describer.txt"))
    for k, n in enumerate(names):
        a, b = int(m.view_off[k]), int(m.view_off[k + 1])
        fileio.write_desc(os.path.join(match_dir, n + ".desc"), m.desc[a:b])
        kp = np.concatenate([m.kpt_xy[a:b], np.full((b - a, 1), 4.8, np.float32), np.zeros((b - a, 1), np.float32)], 1)
        fileio.write_feat(os.path.join(match_dir, n + ".feat"), kp)
        if with_bow is not None:
            fileio.write_mat_bin(os.path.join(match_dir, n + ".bow"), np.asarray(with_bow[k], np.float64).reshape(-1, 1))
    return names


def texture_image(seed, height=480, width=640, n_blobs=400, n_rects=60):
    """A grey test image with structure at several scales (blobs + rectangles), uint8."""
    rng = np.random.Generator(np.random.PCG64(seed))
    img = np.zeros((height, width), np.float64)
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(n_blobs):
        cx, cy = rng.uniform(0, width), rng.uniform(0, height)
        s, a = rng.uniform(2, 18), rng.uniform(-1, 1)
        r = int(4 * s) + 1
        x0, x1 = max(0, int(cx) - r), min(width, int(cx) + r + 1)
        y0, y1 = max(0, int(cy) - r), min(height, int(cy) + r + 1)
        img[y0:y1, x0:x1] += a * np.exp(-((xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2) / (2 * s * s))
    for _ in range(n_rects):
        x0, y0 = int(rng.uniform(0, width - 40)), int(rng.uniform(0, height - 40))
        ww, hh = int(rng.uniform(8, 60)), int(rng.uniform(8, 60))
        img[y0:y0 + hh, x0:x0 + ww] += rng.uniform(-0.8, 0.8)
    img = (img - img.min()) / (img.max() - img.min())
    return (img * 255).astype(np.uint8)


def render_plane_view(texture, px_per_m, R, C, focal, width, height):
    """Pinhole image of the textured plane z = 0 (texture pixel (i, j) covers x = j/px_per_m, y = i/px_per_m)
    seen from a camera with world->camera rotation R and centre C.  Bilinear sampling, border replicated."""
    from scipy import ndimage
    ppx, ppy = width / 2.0, height / 2.0
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    rays_c = np.stack([(u - ppx) / focal, (v - ppy) / focal, np.ones_like(u)], -1)
    rays_w = rays_c @ R                       # R^T applied to each ray (rows of R are camera axes)
    t = -C[2] / rays_w[..., 2]
    X = C[0] + t * rays_w[..., 0]
    Y = C[1] + t * rays_w[..., 1]
    img = ndimage.map_coordinates(texture.astype(np.float64), [Y * px_per_m, X * px_per_m], order=1, mode="nearest")
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def backproject_to_plane(kpt_xy, R, C, focal, width, height):
    """3-D points (z = 0) seen at the given pixels."""
    ppx, ppy = width / 2.0, height / 2.0
    rays_c = np.stack([(kpt_xy[:, 0] - ppx) / focal, (kpt_xy[:, 1] - ppy) / focal, np.ones(len(kpt_xy))], -1)
    rays_w = rays_c @ R
    t = -C[2] / rays_w[:, 2]
    return np.stack([C[0] + t * rays_w[:, 0], C[1] + t * rays_w[:, 1], np.zeros(len(kpt_xy))], -1)


def plane_camera(rng, center_xy, height_m, tilt=0.25):
    """A camera above the plane looking down with a random tilt and roll: (R world->camera, C)."""
    C = np.array([center_xy[0], center_xy[1], height_m])
    target = np.array([center_xy[0] + rng.uniform(-tilt, tilt) * height_m,
                       center_xy[1] + rng.uniform(-tilt, tilt) * height_m, 0.0])
    z = target - C
    z /= np.linalg.norm(z)
    roll = rng.uniform(0, 2 * np.pi)
    up = np.array([np.cos(roll), np.sin(roll), 0.0])
    x = np.cross(up, z)
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    return np.stack([x, y, z]), C


def mldb_like_bank(S, n_images=48, target_rows=400000, nq=2000, device=0, seed0=100):
    """A descriptor bank and a query block with the statistics of real M-LDB descriptors instead of uniform random bits:
    AKAZE + M-LDB extraction (the product's K9, `S` = the sfmlocalization_amd package) over synthetic textured images;
    the query block is the first eight images' descriptors, the bank the remaining images' descriptors replicated with
    a growing number of random bit flips up to `target_rows` rows.  -> (query [nq, 64], bank [rows, 64], stats dict).
    Unrelated M-LDB descriptors sit at 222 +- 42 bits of each other (uniform bits: 243 +- 11), which is what the
    screening bound of the Hamming kernel is sensitive to."""
    ak = S.Akaze(640, 480, device=device)
    descs = []
    for k in range(n_images):
        descs.append(ak.detect_and_compute(texture_image(seed0 + k, 480, 640, n_blobs=900, n_rects=160))[1])
    ak.close()
    q = np.concatenate(descs[:8])[:nq]
    base = np.concatenate(descs[8:])
    reps = max(1, target_rows // max(1, len(base)))
    rng = np.random.Generator(np.random.PCG64(1))
    bank = np.concatenate([flip_bits(rng, base, 6 * r) if r else base for r in range(reps)])
    a = np.unpackbits(q[:256], axis=1).astype(np.int32)
    b = np.unpackbits(base[:2048], axis=1).astype(np.int32)
    dist = (a[:, None, :] != b[None, :, :]).sum(2)
    stats = {"images": n_images, "desc_per_image_mean": float(np.mean([len(d) for d in descs])), "nq": int(len(q)),
             "rows": int(len(bank)), "pair_distance_mean": float(dist.mean()), "pair_distance_std": float(dist.std()),
             "bits_set_mean": float(np.unpackbits(base, axis=1).sum(1).mean())}
    return q, bank, stats
