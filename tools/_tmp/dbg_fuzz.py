import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
import numpy as np, torch
import test_gpu_parity as T
from ant_quantization_amd import _lib
import oracle.antq_oracle as orc
dev = torch.device("cuda:0")
seed = int(sys.argv[1])
def run_case(antq_lib, oracle, dev_, x, alpha, grid, gmax, per_row, ovp, bf16):
    rows, K = x.shape
    plan = _lib.plan_for(grid)
    a_t = torch.from_numpy(np.atleast_1d(alpha).astype(np.float32)).to(dev)
    xin = orc.f32_to_bf16(x) if bf16 else x
    ref, ridx = orc.forward(xin, alpha, grid, gmax, ovp)
    out, idx = _lib.fakequant(T.to_dev(xin, dev, bf16), a_t, plan, gmax, rows, K, per_row, ovp=ovp, want_idx=True)
    out2 = _lib.fakequant(T.to_dev(xin, dev, bf16), a_t, plan, gmax, rows, K, per_row, ovp=ovp)
    if bf16:
        o1 = orc.bf16_to_f32(T.bf16_bits(out)).reshape(-1); o2 = orc.bf16_to_f32(T.bf16_bits(out2)).reshape(-1); rf = orc.bf16_to_f32(ref).reshape(-1)
        xf = orc.bf16_to_f32(xin).reshape(-1)
    else:
        o1 = out.cpu().numpy().reshape(-1); o2 = out2.cpu().numpy().reshape(-1); rf = ref.reshape(-1); xf = x.reshape(-1)
    for nm, o in (("idx-variant", o1), ("noidx-variant", o2)):
        bad = np.nonzero((o.view(np.uint32) != rf.view(np.uint32)) & ~(np.isnan(o) & np.isnan(rf)))[0]
        if bad.size:
            h = plan.host[:80].view(np.uint32)
            print("MISMATCH", nm, "per_row", per_row, "ovp", ovp, "bf16", bf16, "shape", x.shape, "n", bad.size, "kind", plan.is_table, "xdom", h[16] if len(h) > 16 else None)
            print(" grid", repr(grid), "gmax", gmax)
            al = np.atleast_1d(alpha)
            for b in bad[:8]:
                r = b // K if per_row else 0
                a = al[r] if al.size > 1 else al[0]
                s = np.float32(a) / np.float32(gmax)
                print("  i=%d x=%r alpha=%r s=%r d=%r got=%r ref=%r ridx=%d gidx=%d partner x=%r" % (b, xf[b], a, s, np.float32(xf[b]) / s, o[b], rf[b], ridx.reshape(-1)[b], idx.cpu().numpy().reshape(-1)[b], xf[b ^ 1]))
    bi = np.nonzero(idx.cpu().numpy().reshape(-1).astype(np.int32) != ridx.reshape(-1))[0]
    if bi.size: print("IDX MISMATCH n", bi.size, bi[:5])
T.run_case = run_case
T.test_random_grids_fuzz.__wrapped__ if hasattr(T.test_random_grids_fuzz, "__wrapped__") else None
fn = T.test_random_grids_fuzz
getattr(fn, "__wrapped__", fn)(None, None, dev, seed)
