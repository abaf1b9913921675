import torch, sys, time
sys.path.insert(0,'.')
from puzzlenet_amd import ops
dev=torch.device('cuda:0')
g=torch.Generator().manual_seed(3)
xyz=torch.rand(64,2048,3,generator=g).to(dev)
for (B,S) in [(2,512),(16,512),(64,512)]:
    x=xyz[:B].contiguous()
    start=torch.zeros(B,dtype=torch.long,device=dev)
    fi=ops.farthest_point_sample(x,S,start); torch.cuda.synchronize(); print('fps ok',B,flush=True)
    nx=ops.index_points(x,fi); torch.cuda.synchronize()
    t=time.time(); idx=ops.knn(x,nx,32); torch.cuda.synchronize(); print('knn ok',B,S,time.time()-t,int(idx.max()),flush=True)
