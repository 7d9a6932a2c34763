"""Does capturing the engine's forward (15 kernels) in a HIP graph shorten the step?  Eager vs graph replay, same box, same buffers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hse_facerec_tf_amd.tf_inference import AGE_GENDER_PB, TensorFlowInference
B, S = int(os.environ.get("KB_BATCH", 256)), 192
dev = torch.device("cuda:0")
tfi = TensorFlowInference(AGE_GENDER_PB, input_tensor="input_1:0", output_tensor="global_pooling/Mean:0", convert2BGR=True,
                          imageNetUtilsMean=True, input_size=(S, S), max_batch=B, device=0)
eng = tfi.engine
gen = torch.Generator(device=dev); gen.manual_seed(123)
xs = [(torch.rand((B, S, S, 3), device=dev, generator=gen) * 256.0 - 128.0).contiguous() for _ in range(4)]
def timeit(f, n=200):
    for i in range(20): f(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): f(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
eager = lambda i: eng.forward(xs[i & 3])["features"]
s = torch.cuda.Stream()
graphs, outs = [], []
with torch.cuda.stream(s):
    for i in range(4): eager(i)
    torch.cuda.synchronize()
    for i in range(4):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            o = eng.forward(xs[i])["features"]
        graphs.append(g); outs.append(o)
    ref = [eager(i).clone() for i in range(4)]
    for i in range(4):
        graphs[i].replay()
    torch.cuda.synchronize()
    print("graph == eager:", all(bool((outs[i] == ref[i]).all()) for i in range(4)))
    for rep in range(3):
        te = timeit(eager); tg = timeit(lambda i: graphs[i & 3].replay())
        print("eager %.4f ms/step (%.0f faces/s)   graph %.4f ms/step (%.0f faces/s)" % (te, B / te * 1e3, tg, B / tg * 1e3))
