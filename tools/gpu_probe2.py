"""Kernel-time probe: rotating buffers, graph replay, for rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib as L
dev = torch.device("cuda:0")
G = np.load("tests/golden/ant_grids.npz")
g = G["flint_b4_s"]; plan = L.plan_for(g)
for dtype in (torch.bfloat16, torch.float32):
    nb = 32 if dtype == torch.bfloat16 else 16
    xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dtype) for _ in range(nb)]
    outs = [torch.empty_like(xs[0]) for _ in range(nb)]
    alpha = xs[0].float().abs().amax(1).contiguous()
    def step():
        for i in range(nb): L.fakequant(xs[i], alpha, plan, 10.0, 4096, 4096, True, out=outs[i])
    def stepc():
        for i in range(nb): L.copy(xs[i], outs[i])
    for fn, name in ((step, "fakequant"), (stepc, "copy")):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn(); fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                fn()
            gr.replay(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            reps = 20
            for _ in range(reps): gr.replay()
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (reps * nb)
        el = 4096 * 4096; bpe = 2 * xs[0].element_size()
        print("%s %s graph: %.2f us/launch  %.1f Gelem/s  %.2f TB/s" % (name, dtype, ms * 1e3, el / ms / 1e6, el * bpe / ms / 1e9))
    del xs, outs
