import torch, sys
sys.path.insert(0,'.')
from puzzlenet_amd import engine, model5_b, synthetic
import bench
dev=torch.device('cuda:0')
cfg=bench.Cfg(); cfg.num_points=1024
torch.manual_seed(0)
model=model5_b.TouchedRegraster(cfg).to(dev)
batch=synthetic.make_batch(8,1024,dev,seed=1)
torch.manual_seed(5)
r=engine.TrainStep(model,batch,cfg.lr,world=1,use_graph=True,warmup=2)
print('captured; slots',len(r.feed.slots), flush=True)
for i in range(6):
    l=r.step(); torch.cuda.synchronize()
    bad=[n for n,p in model.named_parameters() if not torch.isfinite(p).all()]
    print(i, float(l), 'nonfinite params:', bad[:3], 'lr', r.opt.param_groups[0]['lr'], flush=True)
