import torch, sys
sys.path.insert(0,'.')
from puzzlenet_amd import engine, model5_b, synthetic, distributed as pdist
from puzzlenet_amd import pointnet_util as pu
import bench
dev=torch.device('cuda:0')
cfg=bench.Cfg(); cfg.num_points=1024
torch.manual_seed(0)
model=model5_b.TouchedRegraster(cfg).to(dev)
batch=synthetic.make_batch(8,1024,dev,seed=1)
feed=pu.StartIndexFeed(); pu.set_start_index_feed(feed)
logged={}
model.log=lambda k,v,*a,**kw: logged.__setitem__(k, v)
def body():
    with torch.no_grad():
        return model.training_step(batch,0)['loss']
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(2):
        if i: feed.refill()
        body()
        if i==0: feed.freeze()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
feed.refill()
g=torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out=body()
for i in range(3):
    feed.refill(); g.replay(); torch.cuda.synchronize()
    print(i,float(out), {k: float(v) for k,v in logged.items()}, flush=True)
