import torch, sys
sys.path.insert(0,'.')
from puzzlenet_amd import engine, model5_b, synthetic, distributed as pdist
from puzzlenet_amd import pointnet_util as pu
import bench
mode=sys.argv[1]
dev=torch.device('cuda:0')
cfg=bench.Cfg(); cfg.num_points=1024
torch.manual_seed(0)
model=model5_b.TouchedRegraster(cfg).to(dev)
batch=synthetic.make_batch(8,1024,dev,seed=1)
grads=pdist.FlatGradAllReduce(model.parameters())
feed=pu.StartIndexFeed(); pu.set_start_index_feed(feed)
def body():
    if mode=='fwd':
        with torch.no_grad():
            o=model.predict5(batch,8,need=True,training=True)
        return o[0].sum()
    if mode=='fwd_grad':
        o=model.predict5(batch,8,need=True,training=True)
        return o[0].sum()+o[6].sum()
    if mode=='fwd_bwd_simple':
        grads.zero_()
        o=model.predict5(batch,8,need=True,training=True)
        l=o[0].sum()+o[6].sum()+o[7].sum(); l.backward(); return l
    if mode=='loss_nobwd':
        with torch.no_grad():
            return model.training_step(batch,0)['loss']
    if mode.startswith('mode'):
        cfg.loss_mode=int(mode[4:]); grads.zero_(); l=model.training_step(batch,0)['loss']; l.backward(); return l
    if mode=='full':
        grads.zero_(); l=model.training_step(batch,0)['loss']; l.backward(); return l
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
for i in range(4):
    feed.refill(); g.replay(); torch.cuda.synchronize(); print(mode,i,float(out.detach()),flush=True)
