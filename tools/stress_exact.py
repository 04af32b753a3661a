#!/usr/bin/env python3
"""GPU box: randomised bit-exactness stress of the selection kernels against the CPU oracle - kNN (all slice counts, K, tie
structures), nearest neighbour (Chamfer), cooperative FPS (how many samples a round emits depends on the data; with and without the layout hint).  Not part of the test-suite (minutes of CPU oracle time); run once
after a change to csrc/knn.hip or csrc/patch_ops.hip:   python tools/stress_exact.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu as O, patch_ref as P
from puflow_amd import ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
DEV = "cuda"


def cloud(B, n, kind):
    x = torch.rand(B, n, 3, generator=g) * 2 - 1
    if kind == "lattice":
        s = int(torch.randint(3, 12, (1,), generator=g))
        x = torch.round(x * s) / s
    elif kind == "clusters":
        c = torch.rand(B, 16, 3, generator=g)
        x = c[:, torch.randint(0, 16, (n,), generator=g)] + 1e-3 * torch.rand(B, n, 3, generator=g)
    elif kind == "dupes":
        m = max(n // 7, 1)
        x[:, torch.randint(0, n, (m,), generator=g)] = x[:, :1].clone()
    elif kind == "offset":
        x = x * float(10 ** torch.randint(-3, 2, (1,), generator=g).item()) + (torch.rand(1, 1, 3, generator=g) * 200 - 100)
    elif kind == "surface":
        x = x / x.norm(dim=-1, keepdim=True).clamp_min(1e-6)
    return x.contiguous()


kinds = ["uniform", "lattice", "clusters", "dupes", "offset", "surface"]
t_end = time.time() + budget
n_knn = n_nn1 = n_fps = 0
while time.time() < t_end:
    kind = kinds[int(torch.randint(0, len(kinds), (1,), generator=g))]
    # ---- kNN: (B, N, M) chosen so that all of W = 4 / 8 / 16 and the M < 1024 kernels come up
    B = int(torch.randint(1, 5, (1,), generator=g)); M = int(torch.randint(64, 3000, (1,), generator=g))
    N = int(torch.randint(10, 1500, (1,), generator=g)); K = [4, 8, 16][int(torch.randint(0, 3, (1,), generator=g))]
    pick = int(torch.randint(0, 6, (1,), generator=g))
    if pick == 0:
        B, N = 18, int(torch.randint(3600, 4000, (1,), generator=g))                  # >= 1024 workgroups: the 4-slice variant
        M = int(torch.randint(1024, 1400, (1,), generator=g))
    elif pick == 1:
        B, N = 8, int(torch.randint(3100, 4000, (1,), generator=g))                   # 384 .. 1023 workgroups: 8 slices
        M = int(torch.randint(1024, 1400, (1,), generator=g))
    elif pick == 2:
        B, N = 24, int(torch.randint(180, 700, (1,), generator=g))                    # 64 .. 1024 query tiles and 256 <= M < 1024:
        M = int(torch.randint(256, 1024, (1,), generator=g))                          # knn5_kernel's small-table shapes (training)
    # (picks 0 / 1 now run knn5_kernel - M <= 4096 and >= 64 query tiles; the reference-slice kernels keep the small grids)
    p = cloud(B, M, kind); q = p[:, :N].contiguous() if N <= M and int(torch.randint(0, 2, (1,), generator=g)) else cloud(B, N, kind)
    d_ref, i_ref = O.knn_canonical(q, p, min(K, M))
    i, d = ops.knn_idx32(q.to(DEV), p.to(DEV), min(K, M), want_dist=True)
    assert torch.equal(i.cpu().long(), i_ref) and torch.equal(d.cpu(), d_ref), ("knn", kind, B, N, M, K)
    n_knn += 1
    # ---- nearest neighbour
    Bn = 20 if int(torch.randint(0, 2, (1,), generator=g)) else 2                   # 20 items: >= 64 query tiles -> knn5_kernel<1>
    x, y = cloud(Bn, int(torch.randint(300 if Bn == 20 else 50, 2000, (1,), generator=g)), kind), cloud(Bn, int(torch.randint(260 if Bn == 20 else 33, 3000, (1,), generator=g)), kind)
    d1r, i1r, d2r, i2r = O.chamfer_nn(x, y)
    d1, d2, i1, i2 = ops.chamfer_3DDist()(x.to(DEV), y.to(DEV))
    assert torch.equal(d1.cpu(), d1r) and torch.equal(d2.cpu(), d2r) and torch.equal(i1.cpu().long(), i1r) and torch.equal(i2.cpu().long(), i2r), ("nn1", kind)
    n_nn1 += 1
    # ---- cooperative FPS (N >= 8192)
    n = int(torch.randint(8192, 30000, (1,), generator=g)); m = int(torch.randint(2, 3000, (1,), generator=g))
    c = cloud(int(torch.randint(1, 4, (1,), generator=g)), n, kind)
    ref = P.fps(c, m)
    grp = [0, 768, 1280, 1536][int(torch.randint(0, 4, (1,), generator=g))]         # the merge's layout hint: other points per thread /
    got = ops.furthest_point_sample(c.to(DEV), m, group=grp)                        # workgroups per cloud, the same samples
    assert torch.equal(got.cpu().long(), ref), ("fps", kind, tuple(c.shape), m, grp)
    n_fps += 1
    print(f"ok  kNN {n_knn}  nn1 {n_nn1}  fps {n_fps}   (last: {kind})", flush=True)
print(f"stress passed: {n_knn} kNN, {n_nn1} nearest-neighbour, {n_fps} FPS configurations, all bit-identical to the oracle")
