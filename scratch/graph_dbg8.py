import torch, sys
sys.path.insert(0,'.')
from puzzlenet_amd import engine, model5_b, synthetic, distributed as pdist
from puzzlenet_amd import pointnet_util as pu
import bench
mode=sys.argv[1]; LR=float(sys.argv[2]) if len(sys.argv)>2 else 1e-4
dev=torch.device('cuda:0')
cfg=bench.Cfg(); cfg.num_points=1024
torch.manual_seed(0)
model=model5_b.TouchedRegraster(cfg).to(dev)
batch=synthetic.make_batch(8,1024,dev,seed=1)
grads=pdist.FlatGradAllReduce(model.parameters())
feed=pu.StartIndexFeed(); pu.set_start_index_feed(feed)
opt=torch.optim.SGD(model.parameters(), lr=1e-4)
if mode=='adam': opt=torch.optim.Adam(model.parameters(), lr=1e-4)
if mode.startswith('adam_cap'): opt=torch.optim.Adam(model.parameters(), lr=torch.tensor(LR,device=dev), capturable=True)
if mode=='adam_foreach_off': opt=torch.optim.Adam(model.parameters(), lr=1e-4, foreach=False)
def body():
    grads.zero_(); l=model.training_step(batch,0)['loss']; l.backward(); return l
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(2):
        if i: feed.refill()
        body()
        if i==0: feed.freeze()
        if mode=='adam_cap_pre': opt.step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
feed.refill()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out=body()
gen=torch.Generator(device=dev).manual_seed(1)
for i in range(4):
    feed.refill(); g.replay(); torch.cuda.synchronize()
    gn=grads.flat.norm().item()
    print(mode,i,float(out.detach()),'gradnorm',gn,flush=True)
    if mode=='perturb':
        with torch.no_grad():
            for p in model.parameters(): p.add_(1e-4*torch.randn(p.shape,device=dev,generator=gen))
    elif mode in ('sgd','adam','adam_cap','adam_foreach_off','adam_cap_pre'):
        opt.step()
    elif mode=='perturb_enc_only':
        with torch.no_grad():
            for n,p in model.named_parameters():
                if n.startswith('Encoder'): p.add_(1e-4*torch.randn(p.shape,device=dev,generator=gen))
    elif mode=='perturb_heads_only':
        with torch.no_grad():
            for n,p in model.named_parameters():
                if not n.startswith('Encoder'): p.add_(1e-4*torch.randn(p.shape,device=dev,generator=gen))
    torch.cuda.synchronize()
