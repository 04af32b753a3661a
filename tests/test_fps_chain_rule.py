"""CPU check of the RULE behind `fps_coopm_kernel` (csrc/patch_ops.hip): a round of the cooperative FPS
publishes every wave's KW best points and then runs the sequential algorithm on the published candidates alone, bounded by the
largest of the waves' last published keys.  This numpy restatement of one cloud's rounds (keys = (distance bits, smallest index
first), exactly the kernel's order; same unfused fp32 arithmetic as the oracle) must reproduce the oracle's sample sequence
bit for bit on random clouds, clustered clouds, lattices with thousands of exact ties and clouds with fewer distinct points than
samples - including the two conditions the rule needs (K' >= B; a distance > 0 for every sample after the first of a round)
and the culling test (a sample whose fp32 distance to a wave's bounding box is >= the wave's largest min-distance is skipped)."""
import numpy as np
import pytest
import torch

from oracle import patch_ref as P


def _sqd(p, c):
    d = p - c                                             # fp32, unfused: (dx^2 + dy^2) + dz^2 as in sqd()
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def _rounds(p, npoint, wave_pts, KW, MS, cap):
    """-> (samples, rounds).  Waves = consecutive runs of `wave_pts` points; candidates beyond `cap` make a one-sample round."""
    N = len(p)
    nw = -(-N // wave_pts)
    md = np.full(N, 1e10, np.float32)
    lo = np.stack([p[w * wave_pts:(w + 1) * wave_pts].min(0) for w in range(nw)])
    hi = np.stack([p[w * wave_pts:(w + 1) * wave_pts].max(0) for w in range(nw)])
    dw = np.full(nw, 1e10, np.float32)                    # >= every min-distance of the wave (its last published best)
    pub = [None] * nw
    out = [0]
    last = [0]
    rounds = 0
    key = lambda i: (md[i], -i)                           # larger distance first, then the smaller index
    while len(out) < npoint:
        for w in range(nw):                               # update, with the exact culling test
            sl = slice(w * wave_pts, (w + 1) * wave_pts)
            touched = False
            for c in last:
                q = np.minimum(np.maximum(p[c], lo[w]), hi[w])
                if _sqd(q[None], p[c][None])[0] < dw[w]:
                    md[sl] = np.minimum(md[sl], _sqd(p[sl], p[c]))
                    touched = True
            if touched or pub[w] is None:
                idx = sorted(range(sl.start, min(sl.stop, N)), key=key, reverse=True)[:KW]
                pub[w] = idx
                dw[w] = md[idx[0]]
        lastk = max(key(pw[-1]) for pw in pub if len(pw) == KW) if any(len(pw) == KW for pw in pub) else (-1.0, 0)
        cands = [i for pw in pub for i in pw]
        k1 = max(cands, key=key)
        rel = [i for i in cands if i == k1 or (key(i) >= lastk and md[i] > 0)]
        if len(rel) > cap:
            rel = [k1]
        cmd = {i: md[i] for i in rel}                     # the chain's private copy of the candidates' keys
        ck = lambda i: (cmd[i], -i)
        last = []
        while True:
            c = max(rel, key=ck)
            if last and not (ck(c) >= lastk and cmd[c] > 0):
                break
            last.append(c)
            out.append(c)
            if len(last) == MS or len(out) >= npoint:
                break
            for i in rel:
                cmd[i] = min(cmd[i], _sqd(p[i][None], p[c][None])[0])
        rounds += 1
    return np.array(out[:npoint]), rounds


def _cloud(kind, N, rng):
    if kind == "uniform":
        return rng.random((N, 3), dtype=np.float32)
    if kind == "lattice":                                 # thousands of exact ties, fewer distinct points than samples
        return (np.round(rng.random((N, 3)) * 3) / 3).astype(np.float32)
    if kind == "clusters":
        cen = rng.random((12, 3))
        return (cen[rng.integers(0, 12, N)] + 0.02 * rng.standard_normal((N, 3))).astype(np.float32)
    base = rng.random((N // 4, 3), dtype=np.float32)      # "patches": every point four times, three of them with noise, in runs
    pts = np.repeat(base, 4, axis=0) + np.tile(np.array([0, 1, 1, 1], np.float32), N // 4)[:, None] * 0.01 * rng.standard_normal((N, 3)).astype(np.float32)
    return pts.astype(np.float32)


@pytest.mark.parametrize("kind,N,npoint,wave_pts,KW,MS,cap", [
    ("uniform", 1024, 400, 64, 2, 8, 128), ("uniform", 1536, 500, 128, 4, 64, 128), ("lattice", 1024, 300, 64, 4, 16, 128),
    ("lattice", 640, 200, 64, 2, 64, 8), ("clusters", 1280, 450, 128, 4, 64, 128), ("patches", 1024, 512, 64, 3, 32, 16),
    ("uniform", 1000, 1000, 64, 4, 64, 128)])
def test_rounds_on_published_candidates_reproduce_sequential_fps(kind, N, npoint, wave_pts, KW, MS, cap):
    rng = np.random.default_rng(N * 7 + KW)
    p = _cloud(kind, N, rng)
    ref = P.fps(torch.from_numpy(p)[None], npoint)[0].numpy()
    got, rounds = _rounds(p, npoint, wave_pts, KW, MS, cap)
    assert np.array_equal(got, ref), (kind, int(np.argmax(got != ref)))
    assert rounds < npoint                               # more than one sample per round on average
