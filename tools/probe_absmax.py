"""antq_absmax on 16 x 4096^2 tensors: per row (rows of 4096, groups of 16) and per tensor; time per launch and read bandwidth."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from ant_quantization_amd import _lib
from bench_configs import timed
dev = torch.device("cuda:0")
for dt, esz in ((torch.float32, 4), (torch.bfloat16, 2)):
    xs = [(torch.randn(4096, 4096, device=dev) * 0.02).to(dt) for _ in range(16)]
    for rows, K, per_row in ((4096, 4096, True), (1048576, 16, True), (4096, 4096, False)):
        secs = timed(lambda: [_lib.absmax(x, rows, K, per_row=per_row) for x in xs], 5)
        byt = 16 * 4096 * 4096 * esz
        print("absmax %-9s rows of %5d %s: %6.1f us/launch  %5.2f TB/s (%4.1f%% of 8)" % (str(dt)[6:], K, "per row" if per_row else "per tensor", secs / 16 * 1e6, byt / secs / 1e12, byt / secs / 8e10), flush=True)
# the same kernels on ONE large tensor (16384 x 16384: 1.07 GB fp32 / 0.54 GB bf16): what they do when the launch boundary
# (ramp, tail and -- per tensor -- the 4-byte memset ahead of the reduction) is amortised
for dt, esz in ((torch.float32, 4), (torch.bfloat16, 2)):
    x = (torch.randn(16384, 16384, device=dev) * 0.02).to(dt)
    for rows, K, per_row in ((16384, 16384, True), (16384 * 1024, 16, True), (16384, 16384, False)):
        secs = timed(lambda: _lib.absmax(x, rows, K, per_row=per_row), 5)
        byt = 16384 * 16384 * esz
        print("absmax %-9s 16384^2, rows of %5d %s: %6.1f us/launch  %5.2f TB/s (%4.1f%% of 8)" % (str(dt)[6:], K, "per row" if per_row else "per tensor", secs * 1e6, byt / secs / 1e12, byt / secs / 8e10), flush=True)
    del x
