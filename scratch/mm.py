import torch, time
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev=[torch.cuda.Event(enable_timing=True) for _ in range(it+1)]
    ev[0].record()
    for i in range(it):
        fn(); ev[i+1].record()
    torch.cuda.synchronize()
    ts=sorted(ev[i].elapsed_time(ev[i+1]) for i in range(it))
    return ts[len(ts)//2]*1e3
for m,k,n in ((36864,512,512),(36864,1536,512),(9216,1024,1024),(147456,256,256),(589824,128,128)):
    a=torch.randn(m,k,device='cuda',dtype=torch.float16); b=torch.randn(n,k,device='cuda',dtype=torch.float16)
    us=t(lambda: torch.matmul(a,b.T))
    print(m,k,n,'f16 matmul %.1f us  %.0f TF' % (us, 2.0*m*k*n/us/1e6))
