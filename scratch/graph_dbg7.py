import torch
dev=torch.device('cuda:0')
x=torch.ones(1<<20,device=dev); buf=torch.full((1<<20,),7.0,device=dev); small=torch.full((64,),7.0,device=dev); xs=torch.ones(64,device=dev)
def body():
    buf.zero_(); buf.add_(x)
    small.zero_(); small.add_(xs)
    z=torch.zeros(1<<16,device=dev); z.add_(x[:1<<16])
    c=x.clone(); c.mul_(3)
    e=torch.empty(1<<16,device=dev); e.copy_(x[:1<<16]); e.add_(1)
    return z, c, e
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    z,c,e=body()
for i in range(4):
    g.replay(); torch.cuda.synchronize()
    print(i, buf.min().item(), buf.max().item(), small.min().item(), small.max().item(), z.min().item(), z.max().item(), c.max().item(), e.max().item())
