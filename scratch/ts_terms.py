import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import model_ref as mr
from puzzlenet_amd import model5_b as mb
G=np.load('tests/golden/loss.npz')
dev=torch.device('cuda:0')
cfg=mr.Cfg(loss_mode=0)
m=mb.TouchedRegraster(cfg); mr.fill_params(m)
ref=mr.RefModel(cfg); ref.load_state_dict(m.state_dict())
m.to(dev)
logged={}
m.log=lambda k,v,*a,**kw: logged.__setitem__(k, float(v))
cb=[torch.from_numpy(G[f'ts_batch{i}']) for i in range(8)]
gb=[t.to(dev) for t in cb]
torch.manual_seed(99); loss=m.training_step(gb,0)['loss']
torch.manual_seed(99); rl,terms=ref.training_step(cb,return_terms=True)
print('loss',loss.item(), rl.item(), G['ts0_loss'])
for k,v in terms.items():
    if k in logged: print(k, logged[k], float(v), abs(logged[k]-float(v))/max(abs(float(v)),1e-12))
# boundary index sets
out=m.predict5(gb,4,need=True,training=True)
