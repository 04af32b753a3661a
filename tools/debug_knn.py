import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ref_cpu as O
from puflow_amd import ops
from puflow_amd.weights import synth_patches
p = synth_patches(1, 128, seed=3, surface=False)
p[0, 10] = p[0, 3]; p[0, 50] = p[0, 3]; p[0, 77] = p[0, 76]
p[0, 100:110] = torch.round(p[0, 100:110] * 4) / 4
d_ref, i_ref = O.knn_canonical(p, p, 16)
d, i, _ = ops.knn_points(p.cuda(), p.cuda(), 16)
i = i.cpu(); d = d.cpu()
bad = (i != i_ref).any(-1).nonzero()
print("bad rows", bad.tolist())
for b, n in bad.tolist()[:6]:
    print(n, "gpu", i[b, n].tolist()); print(n, "ref", i_ref[b, n].tolist())
    print("gpu d", d[b, n].tolist()); print("ref d", d_ref[b, n].tolist())
