"""Synthetic IMAGE world for the image-in measurements and tests (bench.py `image_in`, tests/test_gpu_image_in.py).

A large textured ground plane (z = 0) made of square tiles, each tile a "place" with a texture of its own; map views
and query frames are pinhole renderings of the plane from cameras ~10 m above it.  The map's bank is what the product's
own AKAZE + M-LDB extraction (K9) finds in the rendered map views -- every keypoint a landmark at its back-projection
on the plane -- so that the descriptors a query frame is localised with are the ones extracted from it, not planted
copies.  Infrastructure only (like synthdata.py): nothing here is product code; torch is used to draw textures and
render frames on the GPU (on the CPU for small worlds in tests).

Texture model: a shot-noise field -- Gaussian blobs of random sign, amplitude and size (sigma 2.5 ... 9 texels) at
random texel positions, plus random rectangles -- normalised to mean 127.5 and clipped at +-`ksig` standard
deviations.  At 100 texels per metre seen from 10 m with f = 800 px it gives ~2 000 AKAZE keypoints per VGA frame
(threshold 0.001), the feature count BASELINE.json quotes per query.
"""
from dataclasses import dataclass, field

import numpy as np

import synthdata as synth


def make_atlas(seed, tiles, tile_px=1600, device="cpu", blobs_per_tile=16000, rects_per_tile=2500, smin=2.5, smax=9.0,
               ksig=2.5, n_bins=8):
    """-> torch.uint8 [tiles*tile_px, tiles*tile_px] on `device`.  Blob parameters come from NumPy's PCG64 (host), the
    drawing is done with torch: impulses binned by sigma, one Gaussian filter per bin applied in the Fourier domain;
    rectangles through a 2-D difference array and two cumulative sums."""
    import torch
    S = tiles * tile_px
    rng = np.random.Generator(np.random.PCG64(seed))
    n_blobs = blobs_per_tile * tiles * tiles
    cx = rng.integers(0, S, n_blobs)
    cy = rng.integers(0, S, n_blobs)
    sg = rng.uniform(smin, smax, n_blobs)
    amp = rng.uniform(0.3, 1.0, n_blobs) * rng.choice([-1.0, 1.0], n_blobs)
    edges = np.linspace(smin, smax, n_bins + 1)
    # sum over sigma bins of (impulses of the bin) * (Gaussian of the bin's sigma), in the Fourier domain: 8 forward
    # transforms and one inverse (periodic at the atlas border, which no camera sees).  [Not conv2d: MIOpen spends
    # minutes choosing a solver for 1 x 65 filters on a 6400 x 6400 single-channel image.]
    fy = torch.fft.fftfreq(S, device=device, dtype=torch.float32)[:, None]
    fx = torch.fft.rfftfreq(S, device=device, dtype=torch.float32)[None, :]
    f2 = fy * fy + fx * fx
    acc = torch.zeros((S, S // 2 + 1), dtype=torch.complex64, device=device)
    for b in range(n_bins):
        sel = (sg >= edges[b]) & ((sg < edges[b + 1]) if b < n_bins - 1 else (sg <= edges[b + 1]))
        if not sel.any():
            continue
        s = 0.5 * (edges[b] + edges[b + 1])
        imp = torch.zeros((S, S), dtype=torch.float32, device=device)
        idx = torch.from_numpy((cy[sel] * S + cx[sel]).astype(np.int64)).to(device)
        imp.view(-1).index_add_(0, idx, torch.from_numpy(amp[sel].astype(np.float32)).to(device))
        # transfer function of a unit-PEAK Gaussian of standard deviation s: 2 pi s^2 exp(-2 pi^2 s^2 f^2)
        acc += torch.fft.rfft2(imp) * (2 * np.pi * s * s * torch.exp(-2 * np.pi ** 2 * s * s * f2))
        del imp
    img = torch.fft.irfft2(acc, s=(S, S))
    del acc, f2
    n_rects = rects_per_tile * tiles * tiles
    x0 = rng.integers(0, S - 60, n_rects)
    y0 = rng.integers(0, S - 60, n_rects)
    ww = rng.integers(6, 50, n_rects)
    hh = rng.integers(6, 50, n_rects)
    a = rng.uniform(-0.8, 0.8, n_rects).astype(np.float32)
    diff = torch.zeros((S + 1, S + 1), dtype=torch.float32, device=device)
    for (yy, xx, sign) in ((y0, x0, 1.0), (y0, x0 + ww, -1.0), (y0 + hh, x0, -1.0), (y0 + hh, x0 + ww, 1.0)):
        idx = torch.from_numpy((yy * (S + 1) + xx).astype(np.int64)).to(device)
        diff.view(-1).index_add_(0, idx, torch.from_numpy(sign * a).to(device))
    img += diff.cumsum(0).cumsum(1)[:S, :S]
    del diff
    img = (img - img.mean()) / (ksig * img.std())
    return torch.clamp(127.5 + 127.5 * img, 0, 255).to(torch.uint8)


def render(atlas, px_per_m, Rs, Cs, focal, width, height, chunk=32):
    """Pinhole views of the plane z = 0 textured by `atlas` (texel (i, j) covers x = j / px_per_m, y = i / px_per_m):
    world->camera rotations Rs [n, 3, 3], centres Cs [n, 3] -> np.uint8 [n, height, width].  Bilinear, border
    replicated (the geometry of synthdata.render_plane_view, on whatever device the atlas lives on)."""
    import torch
    import torch.nn.functional as F
    dev = atlas.device
    S = atlas.shape[0]
    tex = atlas.to(torch.float32)[None, None]
    ppx, ppy = width / 2.0, height / 2.0
    u, v = torch.meshgrid(torch.arange(width, dtype=torch.float64, device=dev),
                          torch.arange(height, dtype=torch.float64, device=dev), indexing="xy")
    rays_c = torch.stack([(u - ppx) / focal, (v - ppy) / focal, torch.ones_like(u)], -1)      # [H, W, 3]
    out = np.empty((len(Rs), height, width), np.uint8)
    for i0 in range(0, len(Rs), chunk):
        R = torch.from_numpy(np.asarray(Rs[i0:i0 + chunk], np.float64)).to(dev)
        C = torch.from_numpy(np.asarray(Cs[i0:i0 + chunk], np.float64)).to(dev)
        rays_w = torch.einsum("hwc,ncd->nhwd", rays_c, R)                                    # R^T ray per view
        t = -C[:, None, None, 2] / rays_w[..., 2]
        X = (C[:, None, None, 0] + t * rays_w[..., 0]) * px_per_m
        Y = (C[:, None, None, 1] + t * rays_w[..., 1]) * px_per_m
        grid = torch.stack([X / (S - 1) * 2 - 1, Y / (S - 1) * 2 - 1], -1).to(torch.float32)
        img = F.grid_sample(tex.expand(len(R), -1, -1, -1), grid, mode="bilinear", padding_mode="border",
                            align_corners=True)
        out[i0:i0 + len(R)] = torch.clamp(torch.round(img[:, 0]), 0, 255).to(torch.uint8).cpu().numpy()
    return out


@dataclass
class ImageWorld:
    m: object                   # synthdata.SynthMap: the real (extracted) views first, then the padding views
    n_real: int
    frames: np.ndarray          # [n_queries, H, W] u8 query frames
    frame_R: np.ndarray         # true world->camera rotations of the query frames
    frame_C: np.ndarray         # true centres
    frame_place: np.ndarray
    bow: np.ndarray = None      # [n_views, bof_dim] f32 (.bow of every view), when a BoW model was given
    extra: dict = field(default_factory=dict)


def cameras(rng, n, tile_xy, tile_m, height=(9.0, 11.0), tilt=0.12, margin=4.5):
    """n cameras over the tile whose lower-left corner is tile_xy, looking down with a random tilt and roll."""
    Rs, Cs = [], []
    for _ in range(n):
        cxy = (tile_xy[0] + rng.uniform(margin, tile_m - margin), tile_xy[1] + rng.uniform(margin, tile_m - margin))
        R, C = synth.plane_camera(rng, cxy, rng.uniform(*height), tilt=tilt)
        Rs.append(R)
        Cs.append(C)
    return np.stack(Rs), np.stack(Cs)


def build(S, seed, n_real_views, n_queries, tiles=4, tile_px=1600, px_per_m=100.0, focal=800.0, width=640, height=480,
          device=0, torch_device=None, n_pad_views=0, pad_desc_per_view=2000, dense_bow=None, extract_batch=8,
          atlas_kw=None, progress=None, render_slab=256):
    """The world: atlas -> n_real_views rendered map views -> K9 extraction (S = the sfmlocalization_amd package) ->
    SynthMap (one landmark per keypoint), padded with `n_pad_views` views of random descriptors without landmarks;
    n_queries query frames from new cameras over random tiles.  dense_bow: an engine.DenseBow -- every real view's .bow
    vector is then computed from its rendered image by the product's own chain, and the padding views get real views'
    vectors with the visual words permuted (a different permutation per 20 padding views: unrelated "places")."""
    import torch
    tdev = torch_device if torch_device is not None else (torch.device("cuda", device) if torch.cuda.is_available()
                                                          else torch.device("cpu"))
    import time
    t_mark = [time.perf_counter()]

    def lap(what):
        if progress:
            now = time.perf_counter()
            progress(f"{what}: {now - t_mark[0]:.1f} s")
            t_mark[0] = now
    rng = np.random.Generator(np.random.PCG64(seed))
    atlas = make_atlas(seed, tiles, tile_px, tdev, **(atlas_kw or {}))
    lap(f"atlas {tiles}x{tiles} tiles")
    tile_m = tile_px / px_per_m
    n_places = tiles * tiles
    place_xy = np.array([[tile_m * (p % tiles), tile_m * (p // tiles)] for p in range(n_places)])
    view_place = (np.arange(n_real_views) * n_places // max(1, n_real_views)).astype(np.int64)
    Rs, Cs = [], []
    for v in range(n_real_views):
        R, C = cameras(rng, 1, place_xy[view_place[v]], tile_m)
        Rs.append(R[0])
        Cs.append(C[0])
    Rs, Cs = np.stack(Rs), np.stack(Cs)
    exs = [S.Akaze(width, height, device=device) for _ in range(extract_batch)]
    desc_l, kp_l, X_l, off, bow_l = [], [], [], [0], []
    # rendered and extracted a slab of views at a time: 10 000 VGA views are 3 GB of pixels nobody needs afterwards
    for r0 in range(0, n_real_views, render_slab):
        imgs = render(atlas, px_per_m, Rs[r0:r0 + render_slab], Cs[r0:r0 + render_slab], focal, width, height)
        for i0 in range(0, len(imgs), extract_batch):
            chunk = imgs[i0:i0 + extract_batch]
            feats = S.Akaze.detect_and_compute_batch(exs[:len(chunk)], list(chunk)) if len(chunk) > 1 else \
                [exs[0].detect_and_compute(chunk[0])]
            for k, (kp, desc) in enumerate(feats):
                v = r0 + i0 + k
                desc_l.append(desc)
                kp_l.append(kp[:, :2].copy())
                X_l.append(synth.backproject_to_plane(kp[:, :2].astype(np.float64), Rs[v], Cs[v], focal, width, height))
                off.append(off[-1] + len(desc))
                if dense_bow is not None:
                    g = chunk[k]
                    bow_l.append(dense_bow.compute(np.stack([g, g, g], 2)).astype(np.float32))
        del imgs
    for e in exs:
        e.close()
    lap(f"rendered {n_real_views} map views, extracted {off[-1]} descriptors" + (" + their .bow vectors" if dense_bow is not None else ""))
    n_real_rows = off[-1]
    from sfmlocalization_amd import capi
    kpt_real = capi.feat_round_trip(np.concatenate(kp_l).astype(np.float32))   # what the map's .feat files hold
    desc_real = np.concatenate(desc_l)
    # padding views: random descriptors / keypoints, no landmarks
    n_views = n_real_views + n_pad_views
    n_pad_rows = n_pad_views * pad_desc_per_view
    view_off = np.zeros(n_views + 1, np.uint32)
    view_off[:n_real_views + 1] = off
    view_off[n_real_views + 1:] = n_real_rows + pad_desc_per_view * np.arange(1, n_pad_views + 1)
    if n_pad_rows:
        desc = np.concatenate([desc_real, synth.random_descriptors(rng, n_pad_rows)])
        kpt = np.concatenate([kpt_real, np.stack([rng.uniform(0, width, n_pad_rows), rng.uniform(0, height, n_pad_rows)],
                                                 1).astype(np.float32)])
    else:
        desc, kpt = desc_real, kpt_real
    lap("feat round trip + padding rows")
    row_lm = np.full(len(desc), -1, np.int32)
    row_lm[:n_real_rows] = np.arange(n_real_rows, dtype=np.int32)
    m = synth.SynthMap(
        view_id=np.arange(n_views, dtype=np.uint32), view_off=view_off,
        view_wh=np.tile(np.array([[width, height]], np.uint32), (n_views, 1)), desc=desc, kpt_xy=kpt,
        row_landmark=row_lm, landmark_id=np.arange(n_real_rows, dtype=np.uint32) + 1000,
        landmark_X=np.concatenate(X_l), landmark_desc=np.zeros((0, 64), np.uint8),
        landmark_place=np.zeros(0, np.int64), view_place=np.concatenate([view_place, np.full(n_pad_views, -1)]),
        view_R=Rs, view_C=Cs, place_center=np.concatenate([place_xy + tile_m / 2, np.zeros((n_places, 1))], 1),
        intrinsic=(focal, width / 2.0, height / 2.0), width=width, height=height)
    bow = None
    if dense_bow is not None:
        real_bow = np.stack(bow_l)
        K = dense_bow.bof.K
        cells = real_bow.shape[1] // K
        pad = np.empty((n_pad_views, real_bow.shape[1]), np.float32)
        for p0 in range(0, n_pad_views, 20):
            perm = rng.permutation(K)
            src = real_bow[rng.integers(0, n_real_views, min(20, n_pad_views - p0))]
            pad[p0:p0 + len(src)] = src.reshape(len(src), cells, K)[:, :, perm].reshape(len(src), -1)
        bow = np.concatenate([real_bow, pad]) if n_pad_views else real_bow
    # query frames
    # each from a new pose near one of the map views (within a metre of its centre, own height, tilt and roll): the
    # frame overlaps what the map has seen, as a query taken where the map was recorded does
    q_view = rng.integers(0, n_real_views, n_queries)
    q_place = view_place[q_view]
    qR, qC = [], []
    for v in q_view:
        cxy = (Cs[v][0] + rng.uniform(-1.0, 1.0), Cs[v][1] + rng.uniform(-1.0, 1.0))
        R, C = synth.plane_camera(rng, cxy, rng.uniform(9.0, 11.0), tilt=0.10)
        qR.append(R)
        qC.append(C)
    qR, qC = np.stack(qR), np.stack(qC)
    frames = render(atlas, px_per_m, qR, qC, focal, width, height)
    del atlas
    lap("padding .bow + query frames")
    return ImageWorld(m=m, n_real=n_real_views, frames=frames, frame_R=qR, frame_C=qC, frame_place=q_place, bow=bow,
                      extra={"rows_real": n_real_rows, "desc_per_real_view": n_real_rows / max(1, n_real_views)})


def train_bow_model(dense_bow_features, rng, k=100, n_pca=32, iters=4):
    """A BoW model with the reference's shapes (TrainBoW.cpp:46-54: K = 100 words, PCA to 32 dimensions, 2-level
    pyramid, L1-sqrt) from a sample of dense local features [n, 61] f32: PCA by SVD of the centred sample, words by a
    few Lloyd iterations from a random initialisation on the projected (and eigenvalue-divided, PcaWrapper.cpp:67-89)
    sample.  Training is offline and out of scope; this only gives the benchmark a model of the right kind.
    -> (pca dict, bow dict) for fileio.write_cv_yaml."""
    X = np.asarray(dense_bow_features, np.float32)
    mean = X.mean(0, keepdims=True)
    Xc = (X - mean).astype(np.float64)
    cov = Xc.T @ Xc / max(1, len(X) - 1)
    w, V = np.linalg.eigh(cov)
    order = np.argsort(w)[::-1]
    w, V = np.maximum(w[order], 1e-3), V[:, order]
    P = (Xc @ V[:, :n_pca]) / w[None, :n_pca]
    C = P[rng.choice(len(P), k, replace=False)].copy()
    for _ in range(iters):
        d = (P * P).sum(1)[:, None] - 2 * P @ C.T + (C * C).sum(1)[None, :]
        a = d.argmin(1)
        for j in range(k):
            sel = a == j
            if sel.any():
                C[j] = P[sel].mean(0)
    pca = {"DimPCA": n_pca, "EigenVectorsPCA": V.T.astype(np.float32), "EigenValuesPCA": w.reshape(-1, 1).astype(np.float32),
           "MeanPCA": mean.astype(np.float32)}
    bow = {"ResizedImageSize": 300, "UseSpatialPyramid": 1, "PyramidLevel": 2, "NormBofFeatureType": "L1",
           "Centers": C.astype(np.float32)}
    return pca, bow
