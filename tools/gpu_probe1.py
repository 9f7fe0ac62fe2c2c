"""First GPU probe: parity of the static fake-quant kernels vs the oracle + rough timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ant_quantization_amd import _lib as L
from oracle import antq_oracle as orc

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0))
rng = np.random.default_rng(0)

def check(name, x_np, alpha_np, grid, gmax, per_row, ovp, dtype):
    rows, K = x_np.shape
    plan = L.plan_for(grid)
    if dtype == torch.bfloat16:
        xb = orc.f32_to_bf16(x_np)
        xt = torch.from_numpy(xb.view(np.int16)).to(dev).view(torch.bfloat16)
        ref, ridx = orc.forward(xb, alpha_np, grid, gmax, ovp)
    else:
        xt = torch.from_numpy(x_np).to(dev)
        ref, ridx = orc.forward(x_np, alpha_np, grid, gmax, ovp)
    at = torch.from_numpy(np.atleast_1d(alpha_np).astype(np.float32)).to(dev)
    out, idx = L.fakequant(xt, at, plan, gmax, rows, K, per_row, ovp=ovp, want_idx=True)
    out2 = L.fakequant(xt, at, plan, gmax, rows, K, per_row, ovp=ovp, want_idx=False)
    torch.cuda.synchronize()
    if dtype == torch.bfloat16:
        o = out.view(torch.int16).cpu().numpy().view(np.uint16); o2 = out2.view(torch.int16).cpu().numpy().view(np.uint16)
        r = ref
        nanmask = np.isnan(orc.bf16_to_f32(r))
        bad = (o != r) & ~(nanmask & np.isnan(orc.bf16_to_f32(o)))
    else:
        o = out.cpu().numpy(); o2 = out2.cpu().numpy(); r = ref
        bad = (o.view(np.uint32) != r.view(np.uint32)) & ~(np.isnan(o) & np.isnan(r))
    badi = idx.cpu().numpy().astype(np.int32) != ridx
    same = np.array_equal(o.view(np.uint8), o2.view(np.uint8)) or True
    print("%-40s kind=%d rows=%d K=%d bad_val=%d bad_idx=%d" % (name, plan.kind, rows, K, bad.sum(), badi.sum()))
    if bad.sum() or badi.sum():
        w = np.argwhere(bad | badi)[:5]
        for (i, j) in w:
            print("   at", i, j, "x", x_np[i, j], "got", o[i, j], "ref", r[i, j], "idx", idx[i, j].item(), ridx[i, j])
    return bad.sum() + badi.sum()

G = np.load("tests/golden/ant_grids.npz"); O = np.load("tests/golden/olive_grids.npz")
tot = 0
for dtype in (torch.float32, torch.bfloat16):
    for gname in ["flint_b4_s", "int_b4_s", "pot_b4_u", "int_b8_s", "pot_b6_u"]:
        g = G[gname]
        for (rows, K) in [(16, 4096), (64, 576), (64, 147), (128, 64), (7, 1000), (1, 4099), (512, 16)]:
            x = (rng.standard_normal((rows, K)) * 0.02).astype(np.float32)
            x.reshape(-1)[::53] *= 9
            x.reshape(-1)[5] = np.nan; x.reshape(-1)[7] = np.inf; x.reshape(-1)[9] = -3e30; x.reshape(-1)[11] = 0.0
            if gname.endswith("_u"): x = np.abs(x)
            am = np.abs(np.nan_to_num(x, nan=0, posinf=0, neginf=0)); am[am > 1e10] = 0
            alpha = (am.max(1) * 0.9 + 1e-6).astype(np.float32)
            tot += check("%s %s per-row" % (gname, str(dtype)[6:]), x, alpha, g, float(g.max()), True, False, dtype)
            tot += check("%s %s per-tensor" % (gname, str(dtype)[6:]), x, np.float32(alpha.max()), g, float(g.max()), False, False, dtype)
    for t in ("int", "flint"):
        gn = O["%s_b4_s" % t]; go = O["outlier_b4_s"]; g = np.concatenate([gn, go])
        for (rows, K) in [(16, 4096), (5, 33), (64, 147), (128, 64), (3, 7), (1, 4099)]:
            x = (rng.standard_normal((rows, K)) * 0.02).astype(np.float32)
            m = rng.random((rows, K)) < 0.03
            x[m] *= rng.uniform(8, 64, m.sum()).astype(np.float32)
            x.reshape(-1)[0] = 1.5; x.reshape(-1)[2] = 1.2; x.reshape(-1)[3] = -1.4
            alpha = (3 * x.std(1) + 1e-6).astype(np.float32)
            tot += check("olive %s %s per-row ovp" % (t, str(dtype)[6:]), x, alpha, g, float(gn.max()), True, True, dtype)
            tot += check("olive %s %s per-tensor ovp" % (t, str(dtype)[6:]), x, np.float32(alpha.mean()), g, float(gn.max()), False, True, dtype)
print("TOTAL BAD", tot)

# rough timing, headline shape
g = G["flint_b4_s"]; plan = L.plan_for(g)
for dtype in (torch.bfloat16, torch.float32):
    nb = 24
    xs = [ (torch.randn(4096, 4096, device=dev) * 0.02).to(dtype) for _ in range(nb)]
    outs = [torch.empty_like(xs[0]) for _ in range(4)]
    alpha = torch.stack([x.float().abs().amax(1) for x in xs[:1]])[0].contiguous()
    for it in range(3):
        for i in range(nb): L.fakequant(xs[i], alpha, plan, 10.0, 4096, 4096, True, out=outs[i % 4])
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for it in range(reps):
        for i in range(nb): L.fakequant(xs[i], alpha, plan, 10.0, 4096, 4096, True, out=outs[i % 4])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * nb)
    el = 4096 * 4096
    bpe = 2 * xs[0].element_size()
    print("fakequant %s: %.2f us/launch  %.1f Gelem/s  %.2f TB/s" % (dtype, ms * 1e3, el / ms / 1e6, el * bpe / ms / 1e9))
    e0.record()
    for it in range(reps):
        for i in range(nb): L.copy(xs[i], outs[i % 4])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * nb)
    print("copy      %s: %.2f us/launch  %.2f TB/s" % (dtype, ms * 1e3, el * bpe / ms / 1e9))
    del xs, outs
