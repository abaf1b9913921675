import torch, sys, time, cProfile, pstats
sys.path.insert(0,'.')
from puzzlenet_amd import engine, model5_b, synthetic
import bench
dev=torch.device('cuda:0')
cfg=bench.Cfg(); cfg.num_points=2048
torch.manual_seed(0)
model=model5_b.TouchedRegraster(cfg).to(dev)
batch=synthetic.make_batch(64,2048,dev,seed=1)
r=engine.TrainStep(model,batch,cfg.lr,world=1,use_graph=False)
for i in range(3): r.step()
torch.cuda.synchronize()
t0=time.perf_counter()
for i in range(5): r.step()
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print('host issue time/step %.1f ms, wall/step %.1f ms'%((t1-t0)/5*1e3,(t2-t0)/5*1e3))
pr=cProfile.Profile(); pr.enable()
for i in range(5): r.step()
pr.disable(); torch.cuda.synchronize()
st=pstats.Stats(pr); st.sort_stats('tottime').print_stats(25)
