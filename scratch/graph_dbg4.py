import torch, sys
sys.path.insert(0,'.')
from puzzlenet_amd import ops, se3
mode=sys.argv[1]
dev=torch.device('cuda:0')
g0=torch.Generator().manual_seed(0)
a=torch.rand(8,1024,3,generator=g0).to(dev); b=torch.rand(8,1024,3,generator=g0).to(dev)
w=torch.randn(64,128,generator=g0).to(dev).requires_grad_(True); x=torch.randn(4096,128,generator=g0).to(dev)
def body():
    if mode=='emd': return ops.emd_fused(a,b).sum()
    if mode=='chamfer':
        d1,d2=ops.chamfer(a,b); return d1.mean()+d2.mean()
    if mode=='emd3':
        m=ops.emd_approxmatch(a,b); return ops.emd_matchcost(a,b,m).sum()
    if mode=='wgrad':
        w.grad=None
        y=ops.linear(x,w,None,True); y.sum().backward(); return w.grad.sum()
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(2): body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
print(mode,'eager',float(body().detach()))
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out=body()
for i in range(4):
    g.replay(); torch.cuda.synchronize(); print(mode,i,float(out.detach()),flush=True)
