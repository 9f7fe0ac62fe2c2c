"""What the clip-search kernels fetch (1.69 x / 1.95 x the tensor's bytes on a 4096 x 4096 tensor).
launch_search splits the CANDIDATE list over blockIdx.y when a tensor has too few rows to fill the chip (4096 rows = 1024
workgroups -> 2 chunks): every chunk reads the tensor.  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`:
4096 rows (2 chunks) against 8192 and 16384 rows of the same length (1 chunk)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib, core, grids
dev = torch.device("cuda:0")
plans = [_lib.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "pot", "flint")]
for rows in (4096, 8192, 16384):
    for dt in (torch.float32, torch.bfloat16):
        x = (torch.randn(rows, 4096, device=dev) * 0.02).to(dt)
        xm = _lib.absmax(x, rows, 4096)
        ratios = core._ratios(80, 150, 1, dev)
        for _ in range(2):
            _lib.search_sse(x, rows, 4096, xm, True, ratios, plans[2], 10.0)
            _lib.search_sse_multi(x, rows, 4096, xm, True, ratios, plans, [10.0] * 3)
        torch.cuda.synchronize()
        print("rows %5d %s: %d bytes" % (rows, str(dt)[6:], x.numel() * x.element_size()), flush=True)
