import os, sys, time, types
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ant_quantization_amd import _lib, core, grids
from ant_quantization_amd.ant import quant_modules as aq
dev = torch.device("cuda:0")
args = types.SimpleNamespace(mode="flint", wbit=4, abit=4, w_up=150, a_up=150, w_low=75, a_low=75, percent=100, search=False)
lin = aq.LinearQuantizer(mode="flint", wbit=4, abit=4, args=args)
lin.set_param(torch.nn.Linear(64, 64))
lin = lin.to(dev).eval()
x = torch.randn(4, 64, device=dev)
with torch.no_grad():
    lin(x)
    torch.cuda.synchronize()
    for name, fn in (("LinearQuantizer.forward", lambda: lin(x)),
                     ("quant_weight(weight)", lambda: lin.quant_weight(lin.weight, x)),
                     ("core.fake_quant", lambda: core.fake_quant(lin.weight, lin.quant_weight.alpha, lin.quant_weight._plan, 10.0, True)),
                     ("F.linear", lambda: torch.nn.functional.linear(x, lin.weight, lin.bias)),
                     ("torch.empty_like", lambda: torch.empty_like(x))):
        n = 2000
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("%-28s host %.2f us/call   (+sync drain %.2f us/call)" % (name, (t1-t0)/n*1e6, (t2-t1)/n*1e6))
import cProfile, pstats, io
pr = cProfile.Profile()
with torch.no_grad():
    pr.enable()
    for _ in range(2000): lin.quant_weight(lin.weight, x)
    pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
