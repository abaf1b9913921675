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
if mode=='eager_capturable':
    r=engine.TrainStep(model,batch,cfg.lr,world=1,use_graph=False)
    r.opt=torch.optim.Adam(model.parameters(), lr=torch.tensor(cfg.lr,device=dev), capturable=True)
elif mode=='graph_opt_outside':
    r=engine.TrainStep(model,batch,cfg.lr,world=2,use_graph=True,warmup=2)   # world=2 -> optimizer after the graph (no process group: all_reduce is a no-op)
else:
    r=engine.TrainStep(model,batch,cfg.lr,world=1,use_graph=True,warmup=2)
for i in range(5):
    l=r.step(); torch.cuda.synchronize()
    bad=[n for n,p in model.named_parameters() if not torch.isfinite(p).all()]
    badg=[n for n,p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(mode, i, float(l.detach()), 'nonfinite params:', bad[:3], 'grads:', badg[:3], flush=True)
