import torch, sys, os
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from puzzlenet_amd import ops
dev=torch.device('cuda:0')
g=torch.Generator().manual_seed(0)
for (B,N,S) in [(128,2048,512),(128,512,256),(128,1024,512),(128,4096,512),(64,8192,512)]:
    xyz=torch.rand(B,N,3,generator=g).to(dev); st=torch.zeros(B,dtype=torch.long,device=dev)
    for _ in range(3): ops.farthest_point_sample(xyz,S,st)
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): ops.farthest_point_sample(xyz,S,st)
    b.record(); torch.cuda.synchronize()
    print(os.environ.get('PZN_FPS_T','default'),(B,N,S),'%.1f us  %.3f us/iter'%(a.elapsed_time(b)*100, a.elapsed_time(b)*100/S))
