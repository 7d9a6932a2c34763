import torch, time
x = torch.zeros(64, device='cuda')
def run(n):
    for _ in range(n): x.add_(1.0)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run(100); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(2000); e1.record(); torch.cuda.synchronize()
    print('eager tiny kernels: %.2f us each' % (e0.elapsed_time(e1) * 1e3 / 2000))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        run(1000)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print('graph tiny kernels: %.2f us each' % (e0.elapsed_time(e1) * 1e3 / 1000))
