import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import point_ops as orc
from puzzlenet_amd import emd_cuda, ops
dev=torch.device('cuda:0')
B,n,m=2,1024,1024
rng = np.random.default_rng(n * 5 + m)
x1 = rng.random((B, n, 3), dtype=np.float32)
x2 = (x1[:, rng.permutation(n)[:m]] + 0.02 * rng.standard_normal((B, m, 3))).astype(np.float32)
t1=torch.from_numpy(x1).to(dev); t2=torch.from_numpy(x2).to(dev)
c64,m64=orc.earth_mover_distance(x1.astype(np.float64),x2.astype(np.float64))
mg=emd_cuda.approxmatch_forward(t1,t2)
c3=emd_cuda.matchcost_forward(t1,t2,mg).cpu().numpy()
cf=ops.emd_fused(t1,t2).cpu().numpy()
co=orc.emd_matchcost(x1.astype(np.float64),x2.astype(np.float64),mg.cpu().numpy().astype(np.float64))
print('truth',c64,'hip3',c3,'fused',cf,'f64cost(hipmatch)',co)
mgn=mg.cpu().numpy()
print('match rowsum min/max', mgn.sum(1).min(), mgn.sum(1).max(), 'colsum', mgn.sum(2).min(), mgn.sum(2).max(), 'total', mgn.sum((1,2)), m64.sum((1,2)))
print('match maxabs diff', np.abs(mgn-m64).max())
