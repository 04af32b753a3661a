import sys; sys.path.insert(0,'/root/repo')
import torch
from puflow_amd import ops
for (B,N) in ((32,1024),(32,8192)):
    x=torch.rand(B,N,3,device='cuda'); y=torch.rand(B,N,3,device='cuda')
    for _ in range(3): ops.chamfer_nn(x,y)
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.chamfer_nn(x,y)
    b.record(); torch.cuda.synchronize()
    print(B,N,'chamfer fwd ms',a.elapsed_time(b)/10)
