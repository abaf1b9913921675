import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import point_ops as orc
from puzzlenet_amd import emd_cuda, ops
dev=torch.device('cuda:0')
def rel(a,b):
    a=np.asarray(a,np.float64); b=np.asarray(b,np.float64)
    return np.abs(a-b).max()/max(np.abs(b).max(),1e-30)
for (B,n,m) in [(3,128,128),(2,256,128),(2,100,300),(2,1024,1024)]:
    rng=np.random.default_rng(n*3+m)
    x1=rng.random((B,n,3),dtype=np.float32); x2=rng.random((B,m,3),dtype=np.float32)
    t1=torch.from_numpy(x1).to(dev); t2=torch.from_numpy(x2).to(dev)
    mg=emd_cuda.approxmatch_forward(t1,t2).cpu().numpy()
    m32=orc.emd_approxmatch(x1,x2); m64=orc.emd_approxmatch(x1.astype(np.float64),x2.astype(np.float64))
    c32=orc.emd_matchcost(x1,x2,m32); c64=orc.emd_matchcost(x1.astype(np.float64),x2.astype(np.float64),m64)
    cg=emd_cuda.matchcost_forward(t1,t2,torch.from_numpy(mg).to(dev)).cpu().numpy()
    cf=ops.emd_fused(t1,t2).cpu().numpy()
    print((B,n,m),'match: hip-vs-f32 %.2e  f32-vs-f64 %.2e hip-vs-f64 %.2e'%(rel(mg,m32),rel(m32,m64),rel(mg,m64)))
    print('   cost: hip3 %.3e fused %.3e  f32-vs-f64 %.3e'%(rel(cg,c64),rel(cf,c64),rel(c32,c64)), c64)
    gc=np.ones(B,np.float32)
    g1,g2=orc.emd_matchcost_grad(gc.astype(np.float64),x1.astype(np.float64),x2.astype(np.float64),m64)
    o1,o2=orc.emd_matchcost_grad(gc,x1,x2,m32)
    a1=t1.clone().requires_grad_(True); a2=t2.clone().requires_grad_(True)
    ops.emd_fused(a1,a2).sum().backward()
    print('   grad1: fused-vs-f64 %.3e f32-vs-f64 %.3e ; grad2 fused %.3e f32 %.3e'%(rel(a1.grad.cpu().numpy(),g1),rel(o1,g1),rel(a2.grad.cpu().numpy(),g2),rel(o2,g2)))
