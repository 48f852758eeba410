"""Deterministic synthetic camera frames ("Synthetic inputs").

There is no dataset in the build/measurement environment, so every config of BASELINE.json runs on
seeded synthetic grayscale frames: multi-octave value-noise texture + random rectangles (FAST corners)
+ flat regions (exercise the minThFAST fallback and empty cells).  Streams translate by an integer
offset per frame with a small brightness jitter so that consecutive frames match.
"""
import numpy as np

BASE_SEED = 20251004


def _upsample_bilinear(lat, h, w):
    lh, lw = lat.shape
    ys = np.linspace(0, lh - 1, h, dtype=np.float32)
    xs = np.linspace(0, lw - 1, w, dtype=np.float32)
    y0 = np.clip(np.floor(ys).astype(np.int32), 0, lh - 2)
    x0 = np.clip(np.floor(xs).astype(np.int32), 0, lw - 2)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = lat[y0][:, x0]
    b = lat[y0][:, x0 + 1]
    c = lat[y0 + 1][:, x0]
    d = lat[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_world(height, width, seed, n_rects=400, flat_frac=0.15):
    """One large u8 canvas; frames are crops of it."""
    rng = np.random.Generator(np.random.PCG64(seed))
    acc = np.zeros((height, width), np.float32)
    amp = 1.0
    for o in range(6):
        lh, lw = max(2, (height >> (7 - o)) + 2), max(2, (width >> (7 - o)) + 2)
        lat = rng.integers(0, 256, size=(lh, lw)).astype(np.float32)
        acc += amp * _upsample_bilinear(lat, height, width)
        amp *= 0.6
    acc -= acc.min()
    acc *= 255.0 / max(acc.max(), 1e-6)
    img = acc
    # rectangles: axis-aligned and rotated, random gray
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    for i in range(n_rects):
        cx, cy = rng.uniform(0, width), rng.uniform(0, height)
        hw, hh = rng.uniform(4, 40), rng.uniform(4, 40)
        g = rng.uniform(0, 255)
        if i % 2 == 0:
            x0, x1 = int(max(cx - hw, 0)), int(min(cx + hw, width))
            y0, y1 = int(max(cy - hh, 0)), int(min(cy + hh, height))
            img[y0:y1, x0:x1] = g
        else:
            th = rng.uniform(0, np.pi)
            r = int(np.hypot(hw, hh)) + 2
            x0, x1 = int(max(cx - r, 0)), int(min(cx + r, width))
            y0, y1 = int(max(cy - r, 0)), int(min(cy + r, height))
            if x1 <= x0 or y1 <= y0:
                continue
            dx = xx[y0:y1, x0:x1] - cx
            dy = yy[y0:y1, x0:x1] - cy
            u = dx * np.cos(th) + dy * np.sin(th)
            v = -dx * np.sin(th) + dy * np.cos(th)
            m = (np.abs(u) < hw) & (np.abs(v) < hh)
            sub = img[y0:y1, x0:x1]
            sub[m] = g
    # flat regions
    area, target = 0, flat_frac * height * width
    while area < target:
        fw, fh = int(rng.uniform(40, width / 3)), int(rng.uniform(40, height / 3))
        x0, y0 = int(rng.uniform(0, width - fw)), int(rng.uniform(0, height - fh))
        img[y0:y0 + fh, x0:x0 + fw] = rng.uniform(30, 220)
        area += fw * fh
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_frame(height, width, camera_idx=0, frame_idx=0):
    """Independent frame with seed = BASE_SEED + 1000*camera + frame (config 2 / 5 style)."""
    return make_world(height, width, BASE_SEED + 1000 * camera_idx + frame_idx)


def make_stream(height, width, n_frames, camera_idx=0, max_shift=8):
    """Translating stream (config 3): crops of one world + brightness jitter of +-3."""
    seed = BASE_SEED + 1000 * camera_idx
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    margin = max_shift * n_frames + 8
    # keep the rectangle (corner) density of a long stream's world at that of a short one
    n_rects = max(400, int(5e-4 * (height + margin) * (width + margin)))
    world = make_world(height + margin, width + margin, seed, n_rects=n_rects)
    frames = np.empty((n_frames, height, width), np.uint8)
    ox = oy = 0
    for f in range(n_frames):
        crop = world[oy:oy + height, ox:ox + width].astype(np.int16) + int(rng.integers(-3, 4))
        frames[f] = np.clip(crop, 0, 255).astype(np.uint8)
        ox = min(ox + int(rng.integers(0, max_shift + 1)), margin - 1)
        oy = min(oy + int(rng.integers(0, max_shift + 1)), margin - 1)
    return frames


def make_stereo_pair(height, width, camera_idx=0, frame_idx=0, max_disp=60):
    """Left image + right image shifted by a per-row-constant disparity in [5, max_disp] (config 4)."""
    seed = BASE_SEED + 1000 * camera_idx + frame_idx
    rng = np.random.Generator(np.random.PCG64(seed + 13))
    world = make_world(height, width + max_disp, seed)
    left = world[:, max_disp:max_disp + width].copy()
    right = np.empty_like(left)
    # piecewise-constant disparity over bands of rows
    y = 0
    while y < height:
        band = int(rng.integers(8, 48))
        d = int(rng.integers(5, max_disp + 1))
        right[y:y + band] = world[y:y + band, max_disp - d:max_disp - d + width]
        y += band
    return left, right


def random_descriptors(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def planted_descriptors(train, seed, max_flips=60):
    """Each query = a train row with k in [0, max_flips] random bit flips."""
    rng = np.random.Generator(np.random.PCG64(seed))
    q = train.copy()
    for i in range(len(q)):
        k = int(rng.integers(0, max_flips + 1))
        bits = rng.choice(256, size=k, replace=False)
        for b in bits:
            q[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return q


def make_vocabulary(k=10, L=6, seed=7, flips=40, stop_frac=0.02, prune_frac=0.0):
    """Synthetic DBoW2 vocabulary tree in the order of the ORBvoc.txt lines (ORBvoc.txt itself is not in the repo).

    Returns (parent[int32], is_leaf[uint8], desc[n,32 uint8], weight[float64]) for nodes 1..n (node 0 is the root), breadth
    first, so every parent precedes its children.  A child's descriptor is its parent's with ~`flips / level` random bit
    flips (the root's children are random), which makes the greedy descent meaningful; leaves carry an idf-like positive
    weight, `stop_frac` of them weight 0 (stopped words, skipped by transform()); `prune_frac` of the nodes of level L - 1
    have no children and are (shallower) words, as k-means vocabularies have when a cluster runs dry.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    parents, leaves, descs, weights = [], [], [], []
    front_ids = np.zeros(1, np.int64)  # node ids of the inner nodes of the previous level
    front_desc = None
    next_id = 1
    for level in range(1, L + 1):
        m = len(front_ids) * k
        if m == 0:
            break
        pid = np.repeat(front_ids, k)
        if front_desc is None:
            d = rng.integers(0, 256, (m, 32), dtype=np.uint8)
        else:
            bits = np.unpackbits(np.repeat(front_desc, k, axis=0), axis=1)
            nf = max(1, flips // level)
            pos = rng.integers(0, 256, (m, nf))
            np.bitwise_xor.at(bits, (np.repeat(np.arange(m), nf), pos.ravel()), 1)
            d = np.packbits(bits, axis=1)
        leaf = np.ones(m, bool) if level == L else ((rng.random(m) < prune_frac) if level == L - 1 else np.zeros(m, bool))
        w = np.where(leaf, np.where(rng.random(m) < stop_frac, 0.0, rng.uniform(0.5, 12.0, m)), 0.0)
        ids = next_id + np.arange(m)
        next_id += m
        parents.append(pid), leaves.append(leaf), descs.append(d), weights.append(w)
        front_ids, front_desc = ids[~leaf], d[~leaf]
    return (np.concatenate(parents).astype(np.int32), np.concatenate(leaves).astype(np.uint8), np.concatenate(descs).astype(np.uint8),
            np.concatenate(weights).astype(np.float64))


def write_vocabulary_text(path, k, L, parent, is_leaf, desc, weight):
    """ORBvoc.txt format (TemplatedVocabulary::saveToTextFile): header "k L scoring weighting", one node per line."""
    with open(path, "w") as f:
        f.write("%d %d  0 0\n" % (k, L))
        for i in range(len(parent)):
            f.write("%d %d %s %r\n" % (parent[i], is_leaf[i], " ".join(str(int(b)) for b in desc[i]), float(weight[i])))
