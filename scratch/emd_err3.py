import numpy as np, torch, sys
sys.path.insert(0,'.')
from puzzlenet_amd import emd_cuda, ops
dev=torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
B, n = 8, 2048
x1 = torch.rand(B, n, 3, generator=g).to(dev); x2 = torch.rand(B, n, 3, generator=g).to(dev)
match = emd_cuda.approxmatch_forward(x1, x2)
match2 = emd_cuda.approxmatch_forward(x1, x2)
print('approxmatch deterministic:', torch.equal(match,match2))
ones = torch.ones(B, device=dev)
g1, g2 = emd_cuda.matchcost_backward(ones, x1, x2, match)
t1, t2 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
c=ops.emd_fused(t1, t2); c.sum().backward()
t3, t4 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
c2=ops.emd_fused(t3, t4); c2.sum().backward()
print('fused deterministic:', torch.equal(t1.grad,t3.grad), torch.equal(c,c2))
for a,b,name in ((t1.grad,g1,'g1'),(t2.grad,g2,'g2')):
    d=(a-b).abs(); mx=b.abs().max()
    print(name,'maxrel %.2e l2rel %.2e frac>1e-4*max %.4f frac>1e-3*max %.5f'%((d.max()/mx).item(), ((a-b).norm()/b.norm()).item(), (d>1e-4*mx).float().mean().item(), (d>1e-3*mx).float().mean().item()))
cost3=emd_cuda.matchcost_forward(x1,x2,match)
print('cost rel', ((c-cost3).abs()/cost3).max().item(), cost3[:3])
