import numpy as np, sys
def se3_exp(x):
    w=x[:3]; v=x[3:]; t=np.linalg.norm(w)
    K=np.array([[0,-w[2],w[1]],[w[2],0,-w[0]],[-w[1],w[0],0]])
    R=np.eye(3)+np.sin(t)/t*K+(1-np.cos(t))/t**2*K@K
    V=np.eye(3)+(1-np.cos(t))/t**2*K+(t-np.sin(t))/t**3*K@K
    return R, V@v
def sim(x1,x2,label):
    n=len(x1); m=len(x2)
    x1=x1.astype(np.float32); x2=x2.astype(np.float32)
    d=((x1[:,None,:]-x2[None,:,:])**2).sum(-1).astype(np.float32)  # [n,m]
    remL=np.ones(n,np.float32); remR=np.ones(m,np.float32)
    print(label)
    for j in range(7,-3,-1):
        level=np.float32(0 if j==-2 else -4.0**j)
        r=np.sqrt(150/(-level*1.442695)) if level<0 else np.inf
        e=np.exp(level*d).astype(np.float32)
        actR=(remR>0).mean(); actL=(remL>0).mean(); actL8=(remL>1e-6).mean()
        within=(d<r*r)
        # per-row neighbor frac (3D sphere) among active
        fr3=(within[:, remR>0]).mean() if (remR>0).any() else 0
        # linf box
        ratioL=remL/(1e-9+e@remR)
        sumr=remR*(ratioL@e)
        ratioR=np.minimum(remR/(sumr+1e-9),1)*remR
        remR=np.maximum(0,remR-sumr)
        w=e*ratioL[:,None]*ratioR[None,:]
        remL=np.maximum(0,remL-w.sum(1))
        print(f" j={j:2d} r={r:6.3f} actR={actR:.3f} actL>0={actL:.3f} actL>1e-6={actL8:.3f} sphere_frac_of_active={fr3:.4f} massL={remL.sum():8.2f} massR={remR.sum():8.2f}")
if __name__=='__main__':
    rng=np.random.default_rng(0)
    N=int(sys.argv[1]) if len(sys.argv)>1 else 2048
    a=rng.random((N,3)); b=rng.random((N,3))
    sim(a,b,"independent uniform")
    x=rng.standard_normal(6); x=0.8*x/np.linalg.norm(x)
    R,t=se3_exp(x)
    sim(a@R.T+t,a,"rigid moved copy (twist 0.8)")
    x=0.1*x
    R,t=se3_exp(x)
    sim(a@R.T+t,a,"rigid moved copy (twist 0.08)")
