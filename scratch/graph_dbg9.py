import torch, sys
sys.path.insert(0,'.')
from puzzlenet_amd import engine, model5_b, synthetic
import bench
mode=sys.argv[1]
dev=torch.device('cuda:0')
cfg=bench.Cfg(); cfg.num_points=1024
torch.manual_seed(0)
model=model5_b.TouchedRegraster(cfg).to(dev)
batch=synthetic.make_batch(8,1024,dev,seed=1)
torch.manual_seed(5)
if mode=='small_reduce_prewarmed': ok=bool(torch.isfinite(model.dt).all())
r=engine.TrainStep(model,batch,cfg.lr,world=2,use_graph=True,warmup=2)
for i in range(4):
    r.feed.refill(); r.graph.replay()
    if mode=='sched_nosync': r.sched.step()
    if mode=='optsched_nosync': r.opt.step(); r.sched.step()
    torch.cuda.synchronize()
    if mode=='alloc_only':
        tmp=torch.empty(8<<20,device=dev); del tmp
    if mode=='reduce_unrelated':
        big=torch.ones(4<<20,device=dev); ok=bool(torch.isfinite(big).all()); del big
    if mode in ('small_reduce','small_reduce_prewarmed'):
        ok=bool(torch.isfinite(model.dt).all())
    if mode=='isfinite':
        bad=[n for n,p in model.named_parameters() if not torch.isfinite(p).all()]
        badg=[n for n,p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(mode,i,float(r.loss.detach()),'gn',r.grads.flat.norm().item(),flush=True)
    if mode=='opt': r.opt.step()
    if mode=='opt_sched': r.opt.step(); r.sched.step()
    if mode=='none': pass
    torch.cuda.synchronize()
