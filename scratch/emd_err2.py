import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import point_ops as orc
from puzzlenet_amd import emd_cuda, ops
dev=torch.device('cuda:0')
def rel(a,b):
    a=np.asarray(a,np.float64); b=np.asarray(b,np.float64)
    return np.abs(a-b).max()/max(np.abs(b).max(),1e-30)
for (B,n,m) in [(2,512,512),(2,1024,1024)]:
    rng=np.random.default_rng(n*3+m)
    x1=rng.random((B,n,3),dtype=np.float32); x2=rng.random((B,m,3),dtype=np.float32)
    t1=torch.from_numpy(x1).to(dev); t2=torch.from_numpy(x2).to(dev)
    mg=emd_cuda.approxmatch_forward(t1,t2)
    ones=torch.ones(B,device=dev)
    h1,h2=emd_cuda.matchcost_backward(ones,t1,t2,mg)
    a1=t1.clone().requires_grad_(True); a2=t2.clone().requires_grad_(True)
    ops.emd_fused(a1,a2).sum().backward()
    f1=a1.grad.cpu().numpy(); f2=a2.grad.cpu().numpy()
    m64=orc.emd_approxmatch(x1.astype(np.float64),x2.astype(np.float64))
    g1,g2=orc.emd_matchcost_grad(np.ones(B),x1.astype(np.float64),x2.astype(np.float64),m64)
    # oracle grad using HIP match in f64
    q1,q2=orc.emd_matchcost_grad(np.ones(B),x1.astype(np.float64),x2.astype(np.float64),mg.cpu().numpy().astype(np.float64))
    print((B,n,m))
    print(' hip3(hipmatch) vs f64truth: %.2e %.2e'%(rel(h1.cpu().numpy(),g1),rel(h2.cpu().numpy(),g2)))
    print(' f64grad(hipmatch) vs f64truth: %.2e %.2e'%(rel(q1,g1),rel(q2,g2)))
    print(' fused vs f64truth: %.2e %.2e'%(rel(f1,g1),rel(f2,g2)))
    print(' fused vs hip3(hipmatch): %.2e %.2e'%(rel(f1,h1.cpu().numpy()),rel(f2,h2.cpu().numpy())))
    e=np.abs(f1-h1.cpu().numpy()).max(-1); b,k=np.unravel_index(e.argmax(),e.shape); print(' worst g1 at',b,k,f1[b,k],h1[b,k].cpu().numpy(),g1[b,k], 'max|g1|',np.abs(g1).max())
    print(' match row sum for that k:', mg[b,:,k].sum().item(), 'max entry', mg[b,:,k].max().item())
