import torch, sys
import torch.nn.functional as F
sys.path.insert(0,'.')
from puzzlenet_amd import ops, se3
mode=sys.argv[1]
dev=torch.device('cuda:0')
g0=torch.Generator().manual_seed(0)
p=torch.rand(8,1024,generator=g0).to(dev); att=torch.rand(8,256,256,generator=g0).to(dev); x2=torch.rand(8,256,3,generator=g0).to(dev)
tw=torch.randn(8,6,generator=g0).to(dev); pts=torch.rand(8,3,1024,generator=g0).to(dev)
lg=torch.randn(8,2,1024,generator=g0).to(dev); lab=(torch.rand(8,1024,generator=g0)>0.9).float().to(dev)
def body():
    if mode=='topk128': return torch.topk(p,128,1)[1].sum().float()
    if mode=='topk32': return torch.topk(att.mean(dim=1),32)[1][:,0].sum().float()
    if mode=='index':
        idx=torch.topk(att.mean(dim=1),32)[1][:,0]; return x2[:,idx].sum()
    if mode=='se3':
        m=se3.exp(tw); return se3.transform(m,pts).sum()
    if mode=='ce': return F.cross_entropy(lg, lab.squeeze().long())
    if mode=='iou':
        idx=torch.topk(torch.softmax(lg,dim=1)[:,1,:],128,1)[1]
        pr=torch.zeros_like(lab).scatter(1,idx,1)
        return torch.sum(torch.logical_and(pr,lab)).float()/torch.sum(torch.logical_or(pr,lab)).float()
    if mode=='comp':
        m=se3.exp(tw); R=m[:,:3,:3]; t=m[:,:3,3]
        g=torch.eye(4,dtype=R.dtype,device=R.device).unsqueeze(0).repeat(8,1,1); g[:,:3,:3]=R; g[:,:3,3]=t
        A=g.matmul(m); I=torch.eye(4,dtype=A.dtype,device=A.device).view(1,4,4).repeat(8,1,1)
        return F.mse_loss(A,I,reduction='mean')*16
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(2): body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
print(mode,'eager',float(body().detach()))
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out=body()
for i in range(3):
    g.replay(); torch.cuda.synchronize(); print(mode,i,float(out.detach()),flush=True)
