import numpy as np, sys
from emd_sim import se3_exp
def morton(p, lo, hi, bits=10):
    q=np.clip(((p-lo)/(hi-lo+1e-12)*(1<<bits)).astype(np.int64),0,(1<<bits)-1)
    code=np.zeros(len(p),np.int64)
    for b in range(bits):
        for a in range(3):
            code|=((q[:,a]>>b)&1)<<(3*b+a)
    return code
def boxes(p, g):
    n=len(p); k=(n+g-1)//g
    lo=np.full((k,3),np.inf); hi=np.full((k,3),-np.inf)
    for i in range(k):
        s=p[i*g:(i+1)*g]; lo[i]=s.min(0); hi[i]=s.max(0)
    return lo,hi
def frac_pairs(lo1,hi1,lo2,hi2,r, w1, w2):
    gap=np.maximum(0,np.maximum(lo2[None]-hi1[:,None],lo1[:,None]-hi2[None]))
    ok=(gap**2).sum(-1)<=r*r
    return (ok*w1[:,None]*w2[None]).sum()/ (w1.sum()*w2.sum())
def run(x1,x2,label,RG,CH):
    n=len(x1)
    x1=x1.astype(np.float32); x2=x2.astype(np.float32)
    lo=np.minimum(x1.min(0),x2.min(0)); hi=np.maximum(x1.max(0),x2.max(0))
    o1=np.argsort(morton(x1,lo,hi),kind='stable'); o2=np.argsort(morton(x2,lo,hi),kind='stable')
    x1m=x1[o1]; x2m=x2[o2]
    ox1=np.argsort(x1[:,0],kind='stable'); ox2=np.argsort(x2[:,0],kind='stable')
    x1x=x1[ox1]; x2x=x2[ox2]
    d=((x1m[:,None,:]-x2m[None,:,:])**2).sum(-1).astype(np.float32)
    remL=np.ones(n,np.float32); remR=np.ones(n,np.float32)
    print(label, "RG",RG,"CH",CH)
    totB=totCA=totBx=totCAx=totBs=totCAs=0
    for j in range(7,-3,-1):
        level=np.float32(0 if j==-2 else -4.0**j)
        r=np.sqrt(150/(-level*1.442695)) if level<0 else np.inf
        rn=np.sqrt(150/(-level/4*1.442695)) if (level<0 and j>-1) else np.inf
        act=remR>0
        a2=x2m[act]
        # morton boxes: rows=cloud1 groups RG, walked= active list chunks CH   (CA: radius rn ; C alone radius r)
        lo1,hi1=boxes(x1m,RG); lo2,hi2=boxes(a2,CH)
        w1=np.full(len(lo1),RG,float); w2=np.minimum(CH, len(a2)-np.arange(len(lo2))*CH).astype(float)
        fCA=frac_pairs(lo1,hi1,lo2,hi2,rn,w1,w2)*act.mean() if np.isfinite(rn) else act.mean()
        # B: rows = active list groups of RG ; walked = cloud1 chunks CH, radius r
        lo2b,hi2b=boxes(a2,RG); lo1c,hi1c=boxes(x1m,CH)
        w2b=np.minimum(RG, len(a2)-np.arange(len(lo2b))*RG).astype(float); w1c=np.full(len(lo1c),CH,float)
        fB=frac_pairs(lo2b,hi2b,lo1c,hi1c,r,w2b,w1c)*act.mean() if np.isfinite(r) else act.mean()
        # x-window baseline: rows 64, walked quarter-tiles 128 of x sorted (active list order in x)
        # need active in x order: map
        actx=act[np.argsort(o2)][ox2]  # active flags in x order
        a2x=x2x[actx]
        def xfrac(rows, walked, rad, rg=64, ch=128):
            if not np.isfinite(rad): return 1.0
            l1,h1=boxes(rows[:,:1],rg); l2,h2=boxes(walked[:,:1],ch)
            ww1=np.minimum(rg,len(rows)-np.arange(len(l1))*rg).astype(float); ww2=np.minimum(ch,len(walked)-np.arange(len(l2))*ch).astype(float)
            return frac_pairs(l1,h1,l2,h2,rad,ww1,ww2)
        fCAx=xfrac(x1x,a2x,rn)*act.mean(); fBx=xfrac(a2x,x1x,r)*act.mean()
        sphB=(d[:,act]<r*r).mean()*act.mean(); sphCA=(d[:,act]<rn*rn).mean()*act.mean()
        print(f" j={j:2d} act={act.mean():.3f} | B: xwin {fBx:.3f} box {fB:.3f} sphere {sphB:.4f} | CA: xwin {fCAx:.3f} box {fCA:.3f} sphere {sphCA:.4f}")
        totB+=fB; totCA+=fCA; totBx+=fBx; totCAx+=fCAx; totBs+=sphB; totCAs+=sphCA
        e=np.exp(level*d).astype(np.float32)
        ratioL=remL/(1e-9+e@remR); sumr=remR*(ratioL@e)
        ratioR=np.minimum(remR/(sumr+1e-9),1)*remR; remR=np.maximum(0,remR-sumr)
        w=e*ratioL[:,None]*ratioR[None,:]; remL=np.maximum(0,remL-w.sum(1))
    print(f" totals: B xwin {totBx:.2f} box {totB:.2f} sphere {totBs:.2f} | CA xwin {totCAx:.2f} box {totCA:.2f} sphere {totCAs:.2f}")
if __name__=='__main__':
    rng=np.random.default_rng(0)
    N=int(sys.argv[1]); RG=int(sys.argv[2]); CH=int(sys.argv[3])
    a=rng.random((N,3)); b=rng.random((N,3))
    x=rng.standard_normal(6); x=0.8*x/np.linalg.norm(x); R,t=se3_exp(x)
    run(a@R.T+t,a,"rigid 0.8",RG,CH)
    run(a,b,"independent",RG,CH)
