#!/usr/bin/env python3
"""Times pf_fps on the CLI's merge shape (99 840 -> 20 024 points; 1, 8 and 32 clouds at once).
  python tools/time_fps.py [n_points] [n_sample] [n_clouds] [cube|patch]
cube: uniform random points in no order; patch: the merge's shape - 5000 points on a sphere, kNN patches of 256 around random seeds,
every patch point four times with noise, patch after patch (1024 consecutive points = one neighbourhood)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puflow_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 99840
M = int(sys.argv[2]) if len(sys.argv) > 2 else 20024
KIND = sys.argv[4] if len(sys.argv) > 4 else "cube"
GROUP = int(sys.argv[5]) if len(sys.argv) > 5 else 0          # patch: points per patch (1024 = the default synthetic patches)
for B in [int(b) for b in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 8, 32]:
    g = torch.Generator().manual_seed(7)
    if KIND == "patch":
        base = torch.nn.functional.normalize(torch.randn(B, 5000, 3, generator=g), dim=-1).cuda()
        PP = GROUP if GROUP else 1024
        npatch = -(-N // PP)
        seeds = base[:, torch.randperm(5000, generator=g)[:npatch].cuda()]
        nn = torch.cdist(seeds, base).topk(256, largest=False).indices                     # [B, npatch, 256]
        pts = torch.gather(base.unsqueeze(1).expand(B, npatch, 5000, 3), 2, nn.unsqueeze(-1).expand(B, npatch, 256, 3))
        pts = pts.repeat_interleave(PP // 256, dim=2) + 0.01 * torch.randn(B, npatch, PP, 3, generator=g).cuda()
        pc = pts.reshape(B, npatch * PP, 3)[:, :N].contiguous()
    else:
        pc = torch.rand(B, N, 3, generator=g).cuda()
    idx = ops.furthest_point_sample(pc, M, group=GROUP)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 3
    for _ in range(iters):
        idx = ops.furthest_point_sample(pc, M, group=GROUP)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"{KIND} group={GROUP} B={B} N={N} -> {M}: {dt * 1e3:8.2f} ms  ({dt / (M - 1) * 1e6:.3f} us / step)", flush=True)
