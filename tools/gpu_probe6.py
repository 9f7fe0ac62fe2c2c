import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
dev = torch.device("cuda:0")
g = grids.ant_flint(4, True); plan = _lib.plan_for(g)
x = torch.tensor([[1.0, -2.0, 3.0, -4.0, 5.0, -6.0, 7.0, -8.0, 0.5, 0.25, 9.0, 10.0, -10.0, 2.5, 1.25, -0.6]], device=dev)
alpha = torch.tensor([10.0], device=dev)
ref, ridx = _lib.fakequant(x, alpha, plan, 10.0, 1, 16, True, want_idx=True)
codes = _lib.encode4(x, alpha, plan, 10.0, 1, 16, True)
print("ref idx", ridx.tolist()); print("codes", [hex(c) for c in codes.tolist()])
print("dec", _lib.decode4(codes, alpha, plan, 10.0, 1, 16, True, torch.float32).tolist()); print("ref", ref.tolist())
