import torch, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from puzzlenet_amd import _lib
lib=_lib.load()
dev=torch.device('cuda:0')
st=lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, flops, name, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    ms=a.elapsed_time(b)/iters
    print('%-44s %8.3f ms  %6.1f TF/s'%(name, ms, flops/ms/1e9), flush=True)
def P(t): return t.data_ptr() if t is not None else None
g=torch.Generator().manual_seed(0)
def rnd(*s): return torch.randn(*s,device=dev)
for (M,K,N,tag) in [(1048576,128,128,'mlp4'),(524288,256,256,'mlp6'),(1048576,67,128,'mlp3'),(524288,131,256,'mlp5'),(16384,1280,1024,'out'),(131072,64,64,'local'),(16384,256,256,'attn v/out')]:
    x=rnd(M,K); w=rnd(N,K); b=rnd(N); y=torch.empty(M,N,device=dev)
    timeit(lambda: lib.pzn_linear_fwd_f32(P(x),P(w),P(b),M,K,N,1,P(y),st()), 2*M*K*N, f'fwd NT {tag} M={M} K={K} N={N}')
    dy=rnd(M,N); dx=torch.empty(M,K,device=dev)
    timeit(lambda: lib.pzn_linear_dgrad_f32(P(dy),P(y),P(w),M,K,N,None,P(dx),st()), 2*M*K*N, f'dgrad NN {tag} (relu gen)')
    dW=torch.empty(N,K,device=dev); db=torch.empty(N,device=dev)
    timeit(lambda: lib.pzn_linear_wgrad_f32(P(dy),P(y),P(x),M,K,N,P(dW),P(db),0,st()), 2*M*K*N, f'wgrad TN {tag} (relu gen)')
    if M%32==0 and tag in('mlp4','mlp6'):
        R=M//32; out=torch.empty(R,N,device=dev); arg=torch.empty(R,N,dtype=torch.int32,device=dev)
        timeit(lambda: lib.pzn_linear_maxpool_fwd_f32(P(x),P(w),P(b),R,K,N,P(out),P(arg),st()), 2*M*K*N, f'fwd maxpool {tag}')
        dout=rnd(R,N)
        timeit(lambda: lib.pzn_linear_maxpool_dgrad_f32(P(dout),P(arg),P(out),P(w),R,K,N,P(x),P(dx),st()), 2*M*K*N, f'dgrad maxpool-gen+mask {tag}')
        timeit(lambda: lib.pzn_linear_maxpool_wgrad_f32(P(dout),P(arg),P(out),P(x),R,K,N,P(dW),P(db),0,st()), 2*M*K*N, f'wgrad maxpool-gen {tag}')
    del x,y,dy,dx
