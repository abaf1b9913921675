import torch, ctypes, sys
dev=torch.device('cuda:0')
torch.zeros(1,device=dev)
hip=ctypes.CDLL([l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l][0])
hip.hipMemsetAsync.argtypes=[ctypes.c_void_p,ctypes.c_int,ctypes.c_size_t,ctypes.c_void_p]
res={}
for nbytes in (32, 768, 4096, 98304, 1<<20):
    n=nbytes//4
    x=torch.empty(n,device=dev); y=torch.ones(n,device=dev)
    def body():
        x.fill_(7.0)                                   # kernel: garbage
        rc=hip.hipMemsetAsync(x.data_ptr(),0,nbytes,torch.cuda.current_stream().cuda_stream)
        assert rc==0
        x.add_(y)                                      # kernel: accumulate -> expect 1
    s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body()
    out=[]
    for i in range(3):
        g.replay(); torch.cuda.synchronize(); out.append((x.min().item(),x.max().item()))
    print(nbytes,out,flush=True)
