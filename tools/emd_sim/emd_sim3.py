import numpy as np, sys
from emd_sim2 import se3_exp, boxes, frac_pairs
def run(x1,x2,label):
    n=len(x1); x1=x1.astype(np.float32); x2=x2.astype(np.float32)
    ox1=np.argsort(x1[:,0],kind='stable'); ox2=np.argsort(x2[:,0],kind='stable')
    x1x=x1[ox1]; x2x=x2[ox2]
    d=((x1x[:,None,:]-x2x[None,:,:])**2).sum(-1).astype(np.float32)
    remL=np.ones(n,np.float32); remR=np.ones(n,np.float32)
    print(label)
    tot={}
    for j in range(7,-3,-1):
        level=np.float32(0 if j==-2 else -4.0**j)
        r=np.sqrt(150/(-level*1.442695)) if level<0 else np.inf
        rn=np.sqrt(150/(-level/4*1.442695)) if (level<0 and j>-1) else np.inf
        act=remR>0; a2=x2x[act]
        def xfrac(rows, walked, rad, rg, ch):
            if not np.isfinite(rad): return 1.0
            l1,h1=boxes(rows[:,:1],rg); l2,h2=boxes(walked[:,:1],ch)
            ww1=np.minimum(rg,len(rows)-np.arange(len(l1))*rg).astype(float); ww2=np.minimum(ch,len(walked)-np.arange(len(l2))*ch).astype(float)
            return frac_pairs(l1,h1,l2,h2,rad,ww1,ww2)
        out=[]
        for (rg,ch) in [(64,128),(64,32),(64,8),(16,8)]:
            fCA=xfrac(x1x,a2,rn,rg,ch)*act.mean(); fB=xfrac(a2,x1x,r,rg,ch)*act.mean()
            tot[(rg,ch)]=tot.get((rg,ch),0)+fCA+fB
            out.append(f"{fB:.3f}/{fCA:.3f}")
        print(f" j={j:2d} act={act.mean():.3f} B/CA:  "+"   ".join(out))
        e=np.exp(level*d).astype(np.float32)
        ratioL=remL/(1e-9+e@remR); sumr=remR*(ratioL@e)
        ratioR=np.minimum(remR/(sumr+1e-9),1)*remR; remR=np.maximum(0,remR-sumr)
        w=e*ratioL[:,None]*ratioR[None,:]; remL=np.maximum(0,remL-w.sum(1))
    print(" totals (rg,ch):", {k:round(v,2) for k,v in tot.items()})
rng=np.random.default_rng(0); N=2048
a=rng.random((N,3)); b=rng.random((N,3))
x=rng.standard_normal(6); x=0.8*x/np.linalg.norm(x); R,t=se3_exp(x)
run(a@R.T+t,a,"rigid 0.8  (rg,ch)=(64,128) (64,32) (64,8) (16,8)")
run(a,b,"independent")
