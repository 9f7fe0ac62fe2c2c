#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE's own Python.

Runs only in the build container (needs /root/reference); nothing under
tests/ reads /root/reference at test time -- the committed .npz files are the
pins.  Recipe:

  * a stand-in module `quant_cuda` is injected whose quant(x, grid) is the
    literal CPU restatement of ant_quantization/quant/quant_kernel.cu:25-37
    (oracle/antq_oracle.c, via ctypes) and returns (z, zeros_like(x));
  * the reference's antquant/ directory is put first on sys.path and its
    quant_modules.py imported unmodified (one subprocess per tree: both trees
    use the same module names);
  * ANT needs a 1-rank gloo process group (quant_modules.py:525-531).

Usage:  python tests/golden/make_golden.py            (regenerates everything)
"""
import argparse
import os
import subprocess
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402


def scan_numpy(x, grid):
    """A SECOND stand-in for quant_cuda.quant, written independently of oracle/antq_oracle.c and in a different form:
    not the step loop of quant_kernel.cu:29-35 but what that loop computes -- with d_i = fl32|fl32(x) - fl32(y_i)|
    (:23 narrows the grid into `float y_shared[]`, :28 narrows x, :30 subtracts in float), the running `<=` test keeps
    the LAST index attaining the smallest d_i among those with d_i <= 102400 (:25 initial sub_min; a NaN distance
    compares false); z is that grid entry itself, or :26's 0.0 when no entry qualifies.  Every call the reference makes
    while the fixtures are generated is answered by the C scan AND checked against this function (`_install_shim`), so the
    fixtures do not pin the oracle's scan with itself."""
    x = np.asarray(x)
    xf = x.astype(np.float32).reshape(-1)
    y = np.asarray(grid).astype(np.float32).reshape(-1)
    m = y.size
    z = np.zeros(xf.shape, dtype=np.float32)
    idx = np.full(xf.shape, -1, dtype=np.int32)
    for lo in range(0, xf.size, 1 << 16):
        xc = xf[lo:lo + (1 << 16)]
        with np.errstate(all="ignore"):
            d = np.abs(xc[:, None] - y[None, :])                  # float32 throughout
        ok = d <= np.float32(102400.0)                              # False for NaN
        dm = np.where(ok, d, np.float32(np.inf))
        best = dm.min(axis=1)
        last = (m - 1) - np.argmax((dm == best[:, None])[:, ::-1], axis=1)     # last index attaining the minimum
        hit = ok.any(axis=1)
        idx[lo:lo + xc.size] = np.where(hit, last, -1)
        z[lo:lo + xc.size] = np.where(hit, y[np.where(hit, last, 0)], np.float32(0.0))
    return z.astype(x.dtype).reshape(x.shape), idx.reshape(x.shape)


SCAN_CHECKS = [0, 0]      # calls / elements cross-checked between the two stand-ins in this process


def _install_shim():
    import torch
    from oracle import antq_oracle as orc

    shim = types.ModuleType("quant_cuda")
    shim.last_idx = None

    def quant(x, grid):
        xn = x.detach().contiguous().cpu().numpy()
        gn = grid.detach().contiguous().cpu().numpy()
        z, idx = orc.nearest(xn, gn)
        z2, idx2 = scan_numpy(xn, gn)
        zb, z2b = np.ascontiguousarray(z).view(np.uint8), np.ascontiguousarray(z2).view(np.uint8)
        if not (np.array_equal(zb, z2b) and np.array_equal(idx, idx2)):
            raise AssertionError("the two stand-ins for quant_cuda.quant disagree (C scan vs numpy restatement)")
        SCAN_CHECKS[0] += 1
        SCAN_CHECKS[1] += int(xn.size)
        shim.last_idx = idx
        return torch.from_numpy(z).to(x.dtype), torch.zeros_like(x)

    shim.quant = quant
    sys.modules["quant_cuda"] = shim
    return shim


TRACES_ONLY = False      # --traces-only: regenerate nothing but the *_traces.npz files (added in round 2)


def _save(outdir, name, arrays):
    """np.savez_compressed, skipped for the round-1 files under --traces-only (they stay byte-identical)."""
    if TRACES_ONLY and not name.endswith("_traces.npz"):
        return
    np.savez_compressed(os.path.join(outdir, name), **arrays)


# Round 6: beside every recorded float32 score, the same score with the REDUCTION done in float64 over the
# reference's own float32 per-element terms ((q - x).abs().pow(p), AQ:280-285 / OQ:181-187: element-wise fp32, then mean).
# |fp32 score - fp64 score| / score is the reference's own reduction noise: the only thing that may legitimately move a clip
# or type pick between two correct implementations.  Written to separate *_traces64.npz files (the round 1-5
# fixtures stay byte-identical); tests/calib_check.py derives its near-tie tolerance from them.
S64 = []          # parallel to whichever `scores` list is being filled: one float64 vector per mse_loss call


def _score64(qt, st, p, is_perchannel):
    import torch
    e = (qt - st).abs().pow(p)                     # the reference's float32 element terms, bit for bit
    if is_perchannel:
        return e.view(qt.shape[0], -1).double().mean(-1).reshape(-1).numpy()
    return e.double().mean().reshape(-1).numpy()


def _reset(scores):
    del scores[:]
    del S64[:]


def _calibration_trace64(ncand):
    blocks = (len(S64) - 1) // ncand
    return np.stack(S64[(blocks - 1) * ncand:blocks * ncand]).astype(np.float64)


def _put_trace(tr, k, scores, ncand):
    assert len(S64) == len(scores), (len(S64), len(scores))
    tr[k + "__trace"], tr[k + "__type_sums"] = _calibration_trace(scores, ncand)
    tr.setdefault("__64", {})[k + "__trace64"] = _calibration_trace64(ncand)


def _save_traces(outdir, name, tr):
    t64 = tr.pop("__64", {})
    _save(outdir, name, tr)
    np.savez_compressed(os.path.join(outdir, name.replace("_traces.npz", "_traces64.npz")), **t64)


def _calibration_trace(scores, ncand):
    """What a complete `TensorQuantizer(x)` calibration leaves in the mse_loss recorder: for every type of an
    `ant-...` list one search (ncand calls), then the search on the installed grid (ncand calls), then the one call
    for the log value (AQ:519-520 / OQ:286-287).  Returns (final search [ncand, rows], per-type sum of the per-row
    minima [ntypes] float64) -- the numbers the clip pick and the type pick were made from."""
    assert ncand > 0 and (len(scores) - 1) % ncand == 0, (len(scores), ncand)
    blocks = (len(scores) - 1) // ncand
    st = [np.stack(scores[b * ncand:(b + 1) * ncand]).astype(np.float32) for b in range(blocks)]
    sums = np.array([blk.min(axis=0).astype(np.float64).sum() for blk in st[:-1]], dtype=np.float64)
    return st[-1], sums


def _record_mse_loss(qm):
    """Wrap the reference's Quantizer.mse_loss so that every score it returns is appended to the returned list."""
    scores = []
    orig = qm.Quantizer.mse_loss

    def rec_mse(self, qt, st, p=2.0, is_perchannel=True):
        r = orig(self, qt, st, p, is_perchannel)
        scores.append(r.detach().reshape(-1).clone().numpy())
        S64.append(_score64(qt.detach(), st.detach(), p, is_perchannel))
        return r

    qm.Quantizer.mse_loss = rec_mse
    return scores


def _args(**kw):
    d = dict(w_up=150, a_up=150, w_low=75, a_low=75, percent=100, search=False, no_outlier=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def _adversarial_inputs(grid, rng):
    """Values that exercise every rule of the scan for a given grid."""
    g = np.asarray(grid, dtype=np.float32)
    gs = np.unique(g)
    mids = ((gs[:-1].astype(np.float64) + gs[1:].astype(np.float64)) / 2).astype(np.float32)
    around = []
    for v in np.concatenate([gs, mids]):
        around += [v, np.nextafter(v, np.float32(np.inf)), np.nextafter(v, np.float32(-np.inf))]
    special = [0.0, -0.0, 1e-45, -1e-45, 1e-38, -1e-38, np.nan, np.inf, -np.inf,
               1e5, -1e5, 102400.0, -102400.0, 102400.0 + float(gs[-1]), -102400.0 + float(gs[0]),
               102401.0 + float(gs[-1]), -102401.0 + float(gs[0]), 2e5, -2e5, 1e8, -1e8, 3e38, -3e38]
    lo, hi = float(gs[0]), float(gs[-1])
    span = hi - lo
    rnd = np.concatenate([
        rng.standard_normal(1500).astype(np.float32) * np.float32(hi / 3),
        rng.uniform(lo - 0.3 * span, hi + 0.3 * span, 1500).astype(np.float32),
    ])
    return np.concatenate([np.asarray(around, np.float32), np.asarray(special, np.float32), rnd]).astype(np.float32)


# ----------------------------------------------------------------------------
def gen_ant(outdir):
    import torch
    import torch.distributed as dist
    from oracle import antq_oracle as orc

    shim = _install_shim()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    sys.path.insert(0, os.path.join(REF, "ant_quantization", "antquant"))
    import quant_modules as qm

    def mk(mode, bit, signed, is_input=False, **kw):
        q = qm.TensorQuantizer(mode=mode, bit=bit, is_signed=signed, is_enable=True,
                               is_input=is_input, args=_args(**kw))
        q.name = "golden"
        return q

    # ---- (1) every codebook ------------------------------------------------
    grids = {}
    types_ = ["int", "flint", "pot", "float", "float1", "float2", "float3", "float4", "apot"]
    for bit in range(2, 9):
        for signed in (True, False):
            for t in types_:
                q = mk(t, bit, signed)
                key = "%s_b%d_%s" % (t, bit, "s" if signed else "u")
                try:
                    if t == "int":
                        g = q.int_value()
                    elif t == "flint":
                        g = q.flint_value()
                    elif t == "pot":
                        g = q.pot_value()
                    elif t == "apot":
                        g = q.apot_value()
                    elif t == "float":
                        g = q.float_value()
                    else:
                        g = q.float_value(int(t[-1]))
                    grids[key] = g.numpy().astype(np.float32)
                except Exception as e:  # assertion in convert_tensor, etc.
                    grids["INVALID_" + key] = np.array([0], dtype=np.int8)
    _save(outdir, "ant_grids.npz", grids)

    # ---- (2) nearest: literal scan on adversarial inputs -------------------
    rng = np.random.default_rng(20240601)
    near = {}
    for key in ["flint_b4_s", "flint_b4_u", "int_b4_s", "int_b4_u", "pot_b4_s", "pot_b4_u",
                "float_b4_u", "apot_b4_s", "int_b8_s", "int_b8_u", "flint_b6_s", "pot_b6_s", "int_b2_s"]:
        g = grids[key]
        x = _adversarial_inputs(g, rng)
        z, _ = qm.QuantBase.forward(torch.from_numpy(x), torch.from_numpy(g)), None
        near[key + "_x"] = x
        near[key + "_z"] = z[0].numpy() if isinstance(z, tuple) else z.numpy()
        near[key + "_idx"] = shim.last_idx.astype(np.int16)
    # float64 dispatch
    g = grids["flint_b4_s"]
    x64 = _adversarial_inputs(g, rng).astype(np.float64) * (1 + 1e-9)
    z64 = qm.QuantBase.forward(torch.from_numpy(x64), torch.from_numpy(g))
    near["f64_flint_b4_s_x"] = x64
    near["f64_flint_b4_s_z"] = z64.numpy()
    near["f64_flint_b4_s_idx"] = shim.last_idx.astype(np.int16)
    _save(outdir, "ant_nearest.npz", near)

    # ---- (3) _forward with fixed alpha --------------------------------------
    fwd = {}
    torch.manual_seed(3)
    shapes = {"w8x64": (8, 64), "conv1": (64, 3, 7, 7), "w4x768": (4, 768)}
    for sname, shp in shapes.items():
        w = torch.randn(*shp) * 0.05
        w.view(-1)[::97] *= 6.0  # a few clipped values
        fwd[sname + "_x"] = w.numpy()
        for t in ["int", "flint", "pot", "float"]:
            for signed in (True, False):
                x = w if signed else w.abs()
                for per_channel in (True, False):
                    q = mk(t, 4, signed, is_input=not per_channel)
                    q.quant_grid.data = {"int": q.int_value, "flint": q.flint_value,
                                         "pot": q.pot_value, "float": q.float_value}[t]()
                    if per_channel:
                        alpha = x.view(x.shape[0], -1).abs().max(1).values.unsqueeze(1) * 0.9
                    else:
                        alpha = x.abs().max() * 0.8
                    q.alpha.data = alpha
                    with torch.no_grad():
                        out = q._forward(x)
                    k = "%s_%s_%s_%s" % (sname, t, "s" if signed else "u", "pc" if per_channel else "pt")
                    fwd[k + "_alpha"] = alpha.numpy().reshape(-1)
                    fwd[k + "_out"] = out.numpy()
                    fwd[k + "_idx"] = shim.last_idx.reshape(-1).astype(np.int16)
    # 8-bit int per-tensor (configs[0] second leg: TensorQuantizer int8, 256-entry grid)
    torch.manual_seed(0)
    c0 = torch.randn(64, 3, 7, 7) * float(np.sqrt(2.0 / (64 * 49)))
    q = mk("int", 8, True, is_input=True)
    q.quant_grid.data = q.int_value()
    q.alpha.data = c0.abs().max()
    with torch.no_grad():
        out = q._forward(c0)
    fwd["c0_x"] = c0.numpy()
    fwd["c0_int8_pt_alpha"] = q.alpha.data.numpy().reshape(-1)
    fwd["c0_int8_pt_out"] = out.numpy()
    fwd["c0_int8_pt_idx"] = shim.last_idx.reshape(-1).astype(np.int16)
    # group-16 oracle = per-channel reference on x.view(-1, 16)
    torch.manual_seed(8)
    wg = torch.randn(32, 64) * 0.03
    xg = wg.view(-1, 16)
    q = mk("flint", 4, True)
    q.quant_grid.data = q.flint_value()
    q.alpha.data = xg.abs().max(1).values.unsqueeze(1)
    with torch.no_grad():
        out = q._forward(xg)
    fwd["g16_x"] = wg.numpy()
    fwd["g16_flint_alpha"] = q.alpha.data.numpy().reshape(-1)
    fwd["g16_flint_out"] = out.numpy().reshape(32, 64)
    fwd["g16_flint_idx"] = shim.last_idx.reshape(-1).astype(np.int16)
    _save(outdir, "ant_forward.npz", fwd)

    # ---- (5) search_mse traces ----------------------------------------------
    srch, srch64 = {}, {}
    scores = []
    orig_mse = qm.Quantizer.mse_loss

    def rec_mse(self, qt, st, p=2.0, is_perchannel=True):
        r = orig_mse(self, qt, st, p, is_perchannel)
        scores.append(r.detach().reshape(-1).clone().numpy())
        S64.append(_score64(qt.detach(), st.detach(), p, is_perchannel))
        return r

    qm.Quantizer.mse_loss = rec_mse
    torch.manual_seed(5)
    w = torch.randn(16, 256) * 0.02
    a = torch.nn.functional.gelu(torch.randn(8, 512))
    for t in ["int", "flint", "pot"]:
        for name, x, is_input, signed in [("w", w, False, True), ("a", a, True, True), ("au", a.abs(), True, False)]:
            q = mk(t, 4, signed, is_input=is_input)
            q.quant_grid.data = {"int": q.int_value, "flint": q.flint_value, "pot": q.pot_value}[t]()
            _reset(scores)
            with torch.no_grad():
                best, alpha, ratio = q.search_mse(x)
            k = "%s_%s" % (name, t)
            srch[k + "_trace"] = np.stack(scores)
            srch64[k + "_trace64"] = np.stack(S64).astype(np.float64)
            srch[k + "_best_sum"] = np.float32(best.item() if hasattr(best, "item") else best)
            srch[k + "_alpha"] = alpha.numpy().reshape(-1)
            srch[k + "_ratio"] = np.float32(ratio)
    srch["w_x"] = w.numpy()
    srch["a_x"] = a.numpy()
    _save(outdir, "ant_search.npz", srch)

    # ---- (6) full TensorQuantizer: type select + calibration + forward ------
    sel, tr = {}, {}
    torch.manual_seed(7)
    cases = {
        "w_gauss": (torch.randn(32, 128) * 0.02, False),
        "w_unif": ((torch.rand(32, 128) * 2 - 1) * 0.05, False),
        "w_laplace": (torch.distributions.Laplace(0.0, 0.02).sample((32, 128)), False),
        "x_relu": (torch.relu(torch.randn(16, 256)), True),
        "x_gelu": (torch.nn.functional.gelu(torch.randn(16, 256)), True),
    }
    for name, (x, is_input) in cases.items():
        for mode in ["ant-int-pot-flint", "ant-int-flint", "flint", "int"]:
            q = mk(mode, 4, not is_input, is_input=is_input)
            if not is_input:
                q.alpha.data = torch.ones(x.shape[0], 1)
            _reset(scores)
            out = q(x)
            k = "%s__%s" % (name, mode)
            sel[k + "__mode"] = np.array(q.mode)
            sel[k + "__signed"] = np.array(bool(q.is_signed))
            sel[k + "__alpha"] = q.alpha.data.numpy().reshape(-1)
            sel[k + "__grid"] = q.quant_grid.data.numpy()
            sel[k + "__out"] = out.detach().numpy()
            sel[k + "__mse"] = np.float32(q.mse.item())
            sel[k + "__ncand"] = np.int32(len(scores))
            _put_trace(tr, k, scores, 75)
        sel[name + "__x"] = x.numpy()
    # 8-bit forces int (AQ:482-483) with lb=95 (AQ:296-297)
    x = cases["w_gauss"][0]
    q = mk("ant-int-pot-flint", 8, True)
    q.alpha.data = torch.ones(x.shape[0], 1)
    out = q(x)
    sel["w_gauss__b8__mode"] = np.array(q.mode)
    sel["w_gauss__b8__alpha"] = q.alpha.data.numpy().reshape(-1)
    sel["w_gauss__b8__out"] = out.detach().numpy()
    _save(outdir, "ant_select.npz", sel)
    _save_traces(outdir, "ant_select_traces.npz", tr)
    np.savez_compressed(os.path.join(outdir, "ant_search_traces64.npz"), **srch64)
    qm.Quantizer.mse_loss = orig_mse

    # ---- (6b) 'outlier' baseline mode (int4 body + int16 outliers by percentile, AQ:417-465) ----
    outl = {}
    torch.manual_seed(9)
    xo = torch.randn(24, 96) * 0.05
    xo.view(-1)[::41] *= 12
    outl["x"] = xo.numpy()
    for pct in (99.0, 95.0):
        for signed in (True, False):
            xx = xo if signed else xo.abs()
            q = mk("outlier", 4, signed, is_input=not signed, percent=pct)
            out = q(xx)
            k = "p%d_%s" % (int(pct), "s" if signed else "u")
            outl[k + "_out"] = out.detach().numpy()
            outl[k + "_p4"] = np.float32(q.percent_value_int4.item())
            outl[k + "_p16"] = np.float32(q.percent_value_int16.item())
            outl[k + "_out2"] = q(xx * 0.5).detach().numpy()      # second call: steady state on new data
    _save(outdir, "ant_outlier.npz", outl)

    # ---- (7) quant_affine -----------------------------------------------------
    import quant_affine as qa
    aff = {}
    aff["c0_x"] = c0.numpy()
    for k in (4, 8):
        out = qa.AsymmetricQuantFunction.apply(c0, k, c0.min(), c0.max())
        aff["c0_k%d_pt_out" % k] = out.numpy()
        mn = c0.view(64, -1).min(1).values
        mx = c0.view(64, -1).max(1).values
        out = qa.AsymmetricQuantFunction.apply(c0, k, mn, mx)
        aff["c0_k%d_pc_out" % k] = out.numpy()
    torch.manual_seed(11)
    l = torch.randn(16, 48)
    aff["lin_x"] = l.numpy()
    aff["lin_k4_pc_out"] = qa.AsymmetricQuantFunction.apply(l, 4, l.min(1).values, l.max(1).values).numpy()
    aff["lin_k8_pt_out"] = qa.AsymmetricQuantFunction.apply(l, 8, l.min(), l.max()).numpy()
    _save(outdir, "affine.npz", aff)
    dist.destroy_process_group()


# ----------------------------------------------------------------------------
def gen_olive(outdir):
    import torch
    from oracle import antq_oracle as orc

    shim = _install_shim()
    sys.path.insert(0, os.path.join(REF, "olive_quantization", "antquant"))
    import quant_modules as qm

    def mk(mode, bit, signed, is_input=False, **kw):
        kw.setdefault("w_up", 250)
        kw.setdefault("a_up", 250)
        q = qm.TensorQuantizer(mode=mode, bit=bit, is_signed=signed, is_enable=True,
                               is_input=is_input, args=_args(**kw))
        q.name = "golden"
        return q

    grids = {}
    for bit in range(3, 9):
        for signed in (True, False):
            q = mk("int", bit, signed)
            s = "s" if signed else "u"
            for t, fn in (("int", q.int_value), ("flint", q.flint_value), ("outlier", q.outlier_value)):
                try:
                    grids["%s_b%d_%s" % (t, bit, s)] = fn().numpy().astype(np.float32)
                except Exception:
                    grids["INVALID_%s_b%d_%s" % (t, bit, s)] = np.array([0], dtype=np.int8)
    _save(outdir, "olive_grids.npz", grids)

    rng = np.random.default_rng(20240602)
    near = {}
    for t in ("int", "flint"):
        for s in ("s", "u"):
            g = np.concatenate([grids["%s_b4_%s" % (t, s)], grids["outlier_b4_%s" % s]])
            x = _adversarial_inputs(g, rng)
            x = np.concatenate([x, np.float32([40, -40, 36, -36, 33, 32.000004, 31.999998, 400, -400, 1000])])
            z = qm.QuantBase.forward(torch.from_numpy(x), torch.from_numpy(g))
            k = "%s_b4_%s" % (t, s)
            near[k + "_grid"] = g
            near[k + "_x"] = x
            near[k + "_z"] = z.numpy()
            near[k + "_idx"] = shim.last_idx.astype(np.int16)
    _save(outdir, "olive_nearest.npz", near)

    # ---- _forward with OVP ------------------------------------------------------
    fwd = {}
    torch.manual_seed(4)

    def planted(shape, frac=0.02):
        w = torch.randn(*shape) * 0.02
        m = torch.rand(*shape) < frac
        w[m] *= (torch.rand(int(m.sum())) * 56 + 8)
        return w

    tens = {
        "w16x64": planted((16, 64)),
        "w5x33": planted((5, 33), 0.08),       # odd numel: roll wrap-around
        "w3x7": planted((3, 7), 0.3),
        "conv1": planted((64, 3, 7, 7)),
    }
    # force specific pair patterns at the front of w16x64 / w5x33:
    for k in ("w16x64", "w5x33", "w3x7"):
        f = tens[k].view(-1)
        f[0] = 1.5      # element 0 outlier -> victim is 1; with odd numel also the LAST element
        f[1] = 0.3
        f[2] = 1.2      # both outliers: odd one is zeroed
        f[3] = -1.4
        f[4] = 0.01     # even non-outlier, odd outlier -> even zeroed
        f[5] = -2.0
    for name, w in tens.items():
        fwd[name + "_x"] = w.numpy()
        for t in ("int", "flint"):
            for per_channel in (True, False):
                for no_outlier in (False, True):
                    q = mk(t, 4, True, is_input=not per_channel, no_outlier=no_outlier)
                    q.outliers.data = q.outlier_value()
                    q.quant_grid.data = q.int_value() if t == "int" else q.flint_value()
                    w2 = w.view(w.shape[0], -1)
                    if per_channel:
                        mean, std = w2.mean(-1), w2.std(-1)
                        alpha = torch.maximum((mean + 3 * std).abs(), (mean - 3 * std).abs()).unsqueeze(1)
                    else:
                        alpha = torch.maximum((w.mean() + 3 * w.std()).abs(), (w.mean() - 3 * w.std()).abs())
                    q.alpha.data = alpha
                    out = q._forward(w)
                    k = "%s_%s_%s_%s" % (name, t, "pc" if per_channel else "pt", "noout" if no_outlier else "ovp")
                    fwd[k + "_alpha"] = alpha.numpy().reshape(-1)
                    fwd[k + "_out"] = out.numpy()
                    fwd[k + "_idx"] = shim.last_idx.reshape(-1).astype(np.int16)
    # unsigned activations
    torch.manual_seed(14)
    a = torch.relu(planted((6, 50), 0.05))
    fwd["a6x50_x"] = a.numpy()
    for t in ("int", "flint"):
        q = mk(t, 4, False, is_input=True)
        q.outliers.data = q.outlier_value()
        q.quant_grid.data = q.int_value() if t == "int" else q.flint_value()
        q.alpha.data = (a.mean() + 3 * a.std()).abs()
        out = q._forward(a)
        fwd["a6x50_%s_pt_ovp_alpha" % t] = q.alpha.data.numpy().reshape(-1)
        fwd["a6x50_%s_pt_ovp_out" % t] = out.numpy()
        fwd["a6x50_%s_pt_ovp_idx" % t] = shim.last_idx.reshape(-1).astype(np.int16)
    _save(outdir, "olive_forward.npz", fwd)

    # ---- search_mse + full quantiser ---------------------------------------------
    srch, srch64 = {}, {}
    scores = []
    orig_mse = qm.Quantizer.mse_loss

    def rec_mse(self, qt, st, p=2.0, is_perchannel=True):
        r = orig_mse(self, qt, st, p, is_perchannel)
        scores.append(r.detach().reshape(-1).clone().numpy())
        S64.append(_score64(qt.detach(), st.detach(), p, is_perchannel))
        return r

    qm.Quantizer.mse_loss = rec_mse
    torch.manual_seed(6)
    w = planted((16, 256), 0.01)
    a = torch.nn.functional.gelu(planted((8, 512), 0.01) * 40)
    srch["w_x"] = w.numpy()
    srch["a_x"] = a.numpy()
    for t in ("int", "flint"):
        for name, x, is_input in (("w", w, False), ("a", a, True)):
            for no_outlier in (False, True):
                q = mk(t, 4, True, is_input=is_input, no_outlier=no_outlier)
                q.outliers.data = q.outlier_value()
                q.quant_grid.data = q.int_value() if t == "int" else q.flint_value()
                _reset(scores)
                best, alpha, ratio = q.search_mse(x)
                k = "%s_%s_%s" % (name, t, "noout" if no_outlier else "ovp")
                srch[k + "_trace"] = np.stack(scores)
                srch64[k + "_trace64"] = np.stack(S64).astype(np.float64)
                srch[k + "_best_sum"] = np.float32(best.item() if hasattr(best, "item") else best)
                srch[k + "_alpha"] = alpha.numpy().reshape(-1)
                srch[k + "_ratio"] = np.float32(ratio)
    tr = {}
    for name, x, is_input in (("w", w, False), ("a", a, True)):
        for mode in ("ant-int-flint", "flint", "int"):
            q = mk(mode, 4, not is_input, is_input=is_input)
            if not is_input:
                q.alpha.data = torch.ones(x.shape[0], 1)
            _reset(scores)
            out = q(x)
            k = "full_%s__%s" % (name, mode)
            srch[k + "__mode"] = np.array(q.mode)
            srch[k + "__signed"] = np.array(bool(q.is_signed))
            srch[k + "__alpha"] = q.alpha.data.numpy().reshape(-1)
            srch[k + "__grid"] = q.quant_grid.data.numpy()
            srch[k + "__outliers"] = q.outliers.data.numpy()
            srch[k + "__out"] = out.numpy()
            srch[k + "__mse"] = np.float32(q.mse.item())
            srch[k + "__ncand"] = np.int32(len(scores))
            _put_trace(tr, k, scores, len(range(75, 250, 2)))
    qm.Quantizer.mse_loss = orig_mse
    _save(outdir, "olive_search.npz", srch)
    tr.setdefault("__64", {}).update(srch64)
    _save_traces(outdir, "olive_search_traces.npz", tr)


# ----------------------------------------------------------------------------
# Wider end-to-end calibrations (one file of its own so the other fixtures stay byte-identical): the remaining modes
# (pot, float, float1..4, apot, type lists containing them -- incl. the AQ:370-397 quirk that every -floatN search
# runs on float_value(1) while the installed grid is float_value(N)), bit widths 2..7, narrower search windows.
# ----------------------------------------------------------------------------
def gen_ant_wide(outdir):
    import torch
    import torch.distributed as dist

    _install_shim()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
    dist.init_process_group("gloo", rank=0, world_size=1)
    sys.path.insert(0, os.path.join(REF, "ant_quantization", "antquant"))
    import quant_modules as qm

    scores, tr = _record_mse_loss(qm), {}
    torch.manual_seed(21)
    w = torch.distributions.Laplace(0.0, 0.03).sample((24, 96))
    w[::5] *= 0.3
    xa = torch.nn.functional.gelu(torch.randn(8, 192) * 1.5)
    xr = torch.relu(torch.randn(8, 192))
    sel = {"w__x": w.numpy(), "xa__x": xa.numpy(), "xr__x": xr.numpy()}
    combos = [("pot", 4), ("float", 4), ("apot", 4), ("float1", 5), ("float2", 5), ("float3", 6), ("float4", 6),
              ("flint", 3), ("flint", 5), ("flint", 6), ("int", 2), ("int", 3), ("int", 6), ("pot", 3), ("pot", 6),
              ("apot", 6), ("int", 7), ("flint", 7),
              ("ant-int-pot-flint-float", 4), ("ant-float1-float2-flint", 4), ("ant-pot-float-apot", 4),
              ("ant-int-pot-flint", 3), ("ant-int-pot-flint", 5), ("ant-int-flint-float3-apot", 6)]
    keys = []
    for name, x, is_input in (("w", w, False), ("xa", xa, True), ("xr", xr, True)):
        for mode, bit in combos:
            for lo, up in ((75, 150), (90, 110)):
                if (lo, up) != (75, 150) and not mode.startswith("ant-"):
                    continue
                q = qm.TensorQuantizer(mode=mode, bit=bit, is_signed=not is_input, is_enable=True, is_input=is_input,
                                       args=_args(w_low=lo, a_low=lo, w_up=up, a_up=up))
                q.name = "golden"
                if not is_input:
                    q.alpha.data = torch.ones(x.shape[0], 1)
                _reset(scores)
                out = q(x)
                k = "%s__%s__b%d__%d_%d" % (name, mode, bit, lo, up)
                _put_trace(tr, k, scores, up - (95 if bit > 6 else lo))
                keys.append(k)
                sel[k + "__mode"] = np.array(q.mode)
                sel[k + "__signed"] = np.array(bool(q.is_signed))
                sel[k + "__alpha"] = q.alpha.data.numpy().reshape(-1)
                sel[k + "__grid"] = q.quant_grid.data.numpy()
                sel[k + "__out"] = out.detach().numpy()
                sel[k + "__mse"] = np.float32(q.mse.item())
    sel["keys"] = np.array(keys)
    _save(outdir, "ant_select_wide.npz", sel)
    _save_traces(outdir, "ant_select_wide_traces.npz", tr)
    dist.destroy_process_group()



def gen_olive_wide(outdir):
    """OliVe end to end beyond the 4-bit signed cases of olive_search.npz: bit widths 3..8, no_outlier on / off, two
    search windows, weights with planted outliers (per channel, 3-sigma rule) and activations (per tensor), an
    odd-numel activation (the roll wrap-around of OQ:311-320)."""
    import torch

    _install_shim()
    sys.path.insert(0, os.path.join(REF, "olive_quantization", "antquant"))
    import quant_modules as qm

    scores, tr = _record_mse_loss(qm), {}
    torch.manual_seed(31)
    w = torch.randn(24, 96) * 0.02
    m = torch.rand(24, 96) < 0.02
    w[m] *= torch.empty(int(m.sum())).uniform_(6, 40)
    xa = torch.nn.functional.gelu(torch.randn(8, 192) * 1.5)
    xa.view(-1)[::37] *= 9
    xo = torch.randn(7, 33)                    # 231 elements: odd
    xo.view(-1)[::29] *= 15
    sel = {"w__x": w.numpy(), "xa__x": xa.numpy(), "xo__x": xo.numpy()}
    keys = []
    for name, x, is_input in (("w", w, False), ("xa", xa, True), ("xo", xo, True)):
        for mode, bit in (("int", 3), ("flint", 3), ("int", 4), ("flint", 4), ("ant-int-flint", 4), ("int", 5), ("flint", 5),
                          ("ant-int-flint", 5), ("flint", 6), ("ant-int-flint", 6), ("int", 8), ("ant-int-flint", 8)):
            for no_outlier in (False, True):
                for lo, up in ((75, 250), (90, 150)):
                    if (lo, up) != (75, 250) and bit != 4:
                        continue
                    q = qm.TensorQuantizer(mode=mode, bit=bit, is_signed=not is_input, is_enable=True, is_input=is_input,
                                           args=_args(w_low=lo, a_low=lo, w_up=up, a_up=up, no_outlier=no_outlier))
                    q.name = "golden"
                    if not is_input:
                        q.alpha.data = torch.ones(x.shape[0], 1)
                    _reset(scores)
                    out = q(x)
                    k = "%s__%s__b%d__%d_%d__%s" % (name, mode, bit, lo, up, "noout" if no_outlier else "ovp")
                    _put_trace(tr, k, scores, len(range(lo, up, 2)))
                    keys.append(k)
                    sel[k + "__mode"] = np.array(q.mode)
                    sel[k + "__signed"] = np.array(bool(q.is_signed))
                    sel[k + "__alpha"] = q.alpha.data.numpy().reshape(-1)
                    sel[k + "__grid"] = q.quant_grid.data.numpy()
                    sel[k + "__outliers"] = q.outliers.data.numpy()
                    sel[k + "__out"] = out.detach().numpy()
                    sel[k + "__mse"] = np.float32(q.mse.item())
    sel["keys"] = np.array(keys)
    _save(outdir, "olive_select_wide.npz", sel)
    _save_traces(outdir, "olive_select_wide_traces.npz", tr)


# ----------------------------------------------------------------------------
# Rows long enough (1024 fp32 elements = 256 sixteen-byte vectors) for the single-read type selection of the HIP path
# (antq_search_sse_multi): complete calibrations of `ant-...` lists with 2, 3 and 4 candidate types, with their traces.
# ----------------------------------------------------------------------------
def gen_long(outdir, tree):
    import torch

    _install_shim()
    if tree == "ant":
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29535")
        dist.init_process_group("gloo", rank=0, world_size=1)
        sys.path.insert(0, os.path.join(REF, "ant_quantization", "antquant"))
    else:
        sys.path.insert(0, os.path.join(REF, "olive_quantization", "antquant"))
    import quant_modules as qm

    scores, tr, sel, keys = _record_mse_loss(qm), {}, {}, []
    torch.manual_seed(41 if tree == "ant" else 43)
    w = torch.distributions.Laplace(0.0, 0.03).sample((8, 1024))
    w[::3] *= 0.4
    xa = torch.nn.functional.gelu(torch.randn(4, 1024) * 1.5)
    if tree == "olive":
        m = torch.rand(8, 1024) < 0.01
        w[m] *= torch.empty(int(m.sum())).uniform_(6, 40)
        xa.view(-1)[::53] *= 9
    sel.update({"w__x": w.numpy(), "xa__x": xa.numpy()})
    if tree == "ant":
        combos = [("ant-int-pot-flint", 4, {}), ("ant-int-pot-flint-float", 4, {}), ("ant-int-flint", 4, {}),
                  ("ant-float1-float2-flint", 4, {}), ("ant-int-pot-flint", 3, {}), ("ant-int-flint-float3-apot", 5, {})]
        lo, up, step = 75, 150, 1
    else:
        combos = [("ant-int-flint", 4, dict(no_outlier=False)), ("ant-int-flint", 4, dict(no_outlier=True)),
                  ("ant-int-flint", 5, dict(no_outlier=False)), ("ant-int-flint", 3, dict(no_outlier=False))]
        lo, up, step = 75, 250, 2
    for name, x, is_input in (("w", w, False), ("xa", xa, True)):
        for mode, bit, kw in combos:
            q = qm.TensorQuantizer(mode=mode, bit=bit, is_signed=not is_input, is_enable=True, is_input=is_input,
                                   args=_args(w_low=lo, a_low=lo, w_up=up, a_up=up, **kw))
            q.name = "golden"
            if not is_input:
                q.alpha.data = torch.ones(x.shape[0], 1)
            _reset(scores)
            out = q(x)
            k = "%s__%s__b%d__%d_%d" % (name, mode, bit, lo, up)
            if tree == "olive":
                k += "__noout" if kw["no_outlier"] else "__ovp"
                sel[k + "__outliers"] = q.outliers.data.numpy()
            keys.append(k)
            sel[k + "__mode"] = np.array(q.mode)
            sel[k + "__signed"] = np.array(bool(q.is_signed))
            sel[k + "__alpha"] = q.alpha.data.numpy().reshape(-1)
            sel[k + "__grid"] = q.quant_grid.data.numpy()
            sel[k + "__out"] = out.detach().numpy()
            sel[k + "__mse"] = np.float32(q.mse.item())
            _put_trace(tr, k, scores, len(range(lo, up, step)))
    sel["keys"] = np.array(keys)
    np.savez_compressed(os.path.join(outdir, "%s_select_long.npz" % tree), **sel)
    t64 = tr.pop("__64", {})
    np.savez_compressed(os.path.join(outdir, "%s_select_long_traces.npz" % tree), **tr)
    np.savez_compressed(os.path.join(outdir, "%s_select_long_traces64.npz" % tree), **t64)


# ----------------------------------------------------------------------------
# Round 3: CHECKPOINTS WRITTEN BY THE REFERENCE'S OWN CODE and one MultiheadAttentionQuantizer forward (N4).
# The reference's quant_model.py / quant_utils.py are imported unmodified (torchvision is absent from the image and is
# only used by get_model(): an empty stand-in module satisfies the import), a small network is rewritten by ITS
# quantize_model, calibrated by ITS first forward, and ITS state_dict() is stored the way ImageNet/main.py saves it from a
# DistributedDataParallel model (every key prefixed "module.", main.py:151-157 strips 7 characters on load).  Every
# TensorQuantizer's (input, output) of the recorded forward is stored beside it: the quantisers are bit-exact across
# devices, F.conv2d / F.linear are not, so the wire test compares quantiser by quantiser and the model output with a
# tolerance.
# ----------------------------------------------------------------------------
def _stub_torchvision():
    import importlib.machinery
    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
    tv.__spec__ = importlib.machinery.ModuleSpec("torchvision", None)      # (transformers probes find_spec("torchvision"))
    tvm.__spec__ = importlib.machinery.ModuleSpec("torchvision.models", None)
    tv.models = tvm
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, tvm


def _record_model(prefix, model, x, fx, qm, call=None, sd="all"):
    """Run model(x) with a hook on every TensorQuantizer; store state_dict (DDP-prefixed), x, y and the hooks' tensors.
    sd: "all", "quant" (only the quantisers' entries: the weights are those of an earlier record) or None.  A weight
    quantiser's input is the layer's weight (in the state dict), so only activation inputs are stored."""
    import torch
    recs, hooks = {}, []
    for name, mod in model.named_modules():
        if isinstance(mod, qm.TensorQuantizer):
            def hook(m, inp, out, name=name):
                recs[name] = (inp[0].detach().clone().numpy(), out.detach().clone().numpy())
            hooks.append(mod.register_forward_hook(hook))
    with torch.no_grad():
        y = call(model, x) if call else model(x)
    for h in hooks:
        h.remove()
    fx[prefix + "x"] = x.numpy()
    fx[prefix + "y"] = y.detach().numpy()
    for k, v in (model.state_dict().items() if sd else ()):
        if sd == "all" or "quant_" in k:
            fx[prefix + "sd__module." + k] = v.detach().numpy()
    for name, (i, o) in recs.items():
        if "quant_input" in name:
            fx[prefix + "q__" + name + "__in"] = i
        fx[prefix + "q__" + name + "__out"] = o
    fx[prefix + "quantizers"] = np.array(sorted(recs.keys()))


def gen_ckpt(outdir, tree):
    import torch
    import torch.nn as nn

    _install_shim()
    _stub_torchvision()
    fx = {}
    if tree == "ant":
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29536")
        dist.init_process_group("gloo", rank=0, world_size=1)
        sys.path.insert(0, os.path.join(REF, "ant_quantization", "antquant"))
        import quant_modules as qm
        import quant_model
        import quant_utils
        args = _args(mode="ant-int-pot-flint", wbit=4, abit=4)
        quant_utils.set_quantizer(args)
        torch.manual_seed(51)
        net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Flatten(), nn.Linear(512, 32), nn.ReLU(),
                            nn.Linear(32, 10))
        x = torch.randn(4, 3, 8, 8)
        model = quant_model.quantize_model(net)
        quant_utils.enable_quantization(model)
        _record_model("a__", model, x, fx, qm)            # first forward = calibration; the state dict is the calibrated one
        x2 = torch.randn(4, 3, 8, 8)
        _record_model("a2__", model, x2, fx, qm, sd=None)          # steady state on new data, same state dict
        # mixed precision the reference's way: pair 1 (the 512 -> 32 Linear) to 8 bit, every quantiser re-armed,
        # the next forward recalibrates -- that layer's quant_grid is 256 entries long in the checkpoint
        quant_model.set_8_bit_layer_l(model, "1")
        _record_model("b__", model, x, fx, qm, sd="quant")
        # N4: one MultiheadAttentionQuantizer (multihead_attention.py:486-686), (seq, batch, embed) and batch_first
        for tag, bf in (("m__", False), ("mb__", True)):
            torch.manual_seed(52)
            ma = nn.Sequential(nn.MultiheadAttention(64, 4, batch_first=bf)).eval()
            qma = quant_model.quantize_model(ma).eval()
            quant_utils.enable_quantization(qma)
            xm = torch.randn(3, 10, 64) if bf else torch.randn(10, 3, 64)
            _record_model(tag, qma, xm, fx, qm, call=lambda m, t: m[0](t, t, t)[0])
            with torch.no_grad():
                fx[tag + "attn_weights"] = qma[0](xm, xm, xm)[1].numpy()
        dist.destroy_process_group()
    else:
        sys.path.insert(0, os.path.join(REF, "olive_quantization", "antquant"))
        import quant_modules as qm
        import quant_model
        import quant_utils
        from transformers import pytorch_utils
        args = _args(mode="ant-int-flint", wbit=4, abit=4, w_up=250, a_up=250)
        quant_utils.set_quantizer(args)
        torch.manual_seed(53)
        net = nn.Sequential(nn.Linear(64, 128), nn.GELU(), nn.Linear(128, 64), pytorch_utils.Conv1D(32, 64))
        with torch.no_grad():
            for m in net:
                if hasattr(m, "weight"):
                    w = m.weight
                    mask = torch.rand_like(w) < 0.01
                    w[mask] *= torch.empty(int(mask.sum())).uniform_(8, 40)
        x = torch.randn(8, 64)
        x.view(-1)[::37] *= 12
        model = quant_model.quantize_model(net)
        quant_utils.enable_quantization(model)
        _record_model("a__", model, x, fx, qm)
        x2 = torch.randn(8, 64)
        x2.view(-1)[::29] *= 10
        _record_model("a2__", model, x2, fx, qm, sd=None)
    np.savez_compressed(os.path.join(outdir, "%s_ckpt.npz" % tree), **fx)


def gen_mha_options(outdir):
    """Round 5: the reference's MultiheadAttentionQuantizer with the options of torch's attention it forwards to the vendored
    `multi_head_attention_forward` (multihead_attention.py:543-547, :598-606, :677-679): add_bias_kv, add_zero_attn, and both
    together with batch_first.  (What cannot be recorded: distinct kdim / vdim -- the reference's set_param raises NameError
    on them, :566-569 -- and either option together with an attention / key-padding mask: its vendored forward calls an
    undefined `pad`, :382.)  Rewritten, calibrated and run by the reference
    itself; stored like the `m__` / `mb__` records of ant_ckpt.npz."""
    import torch
    import torch.nn as nn
    import torch.distributed as dist
    _install_shim()
    _stub_torchvision()
    fx = {}
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29537")
    dist.init_process_group("gloo", rank=0, world_size=1)
    sys.path.insert(0, os.path.join(REF, "ant_quantization", "antquant"))
    import quant_modules as qm
    import quant_model
    import quant_utils
    quant_utils.set_quantizer(_args(mode="ant-int-pot-flint", wbit=4, abit=4))
    for tag, kw, masks in (("mk__", dict(add_bias_kv=True), False), ("mz__", dict(add_zero_attn=True), False),
                           ("mkz__", dict(add_bias_kv=True, add_zero_attn=True, batch_first=True), False)):
        torch.manual_seed(54)
        ma = nn.Sequential(nn.MultiheadAttention(64, 4, **kw)).eval()
        qma = quant_model.quantize_model(ma).eval()
        quant_utils.enable_quantization(qma)
        bf = kw.get("batch_first", False)
        xm = torch.randn(3, 10, 64) if bf else torch.randn(10, 3, 64)
        extra = {}
        if masks:
            kpm = torch.zeros(3, 10, dtype=torch.bool)
            kpm[1, 7:] = True
            kpm[2, 9:] = True
            am = torch.randn(10, 10) * 0.5
            extra = dict(key_padding_mask=kpm, attn_mask=am)
            fx[tag + "key_padding_mask"] = kpm.numpy()
            fx[tag + "attn_mask"] = am.numpy()
        _record_model(tag, qma, xm, fx, qm, call=lambda m, t: m[0](t, t, t, **extra)[0])
        with torch.no_grad():
            fx[tag + "attn_weights"] = qma[0](xm, xm, xm, **extra)[1].numpy()
    dist.destroy_process_group()
    np.savez_compressed(os.path.join(outdir, "ant_mha_options.npz"), **fx)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree", choices=["ant", "ant_wide", "olive", "olive_wide", "ant_long", "olive_long", "ant_ckpt", "olive_ckpt", "ant_mha", "all"], default="all")
    ap.add_argument("--out", default=HERE)
    ap.add_argument("--traces-only", action="store_true",
                    help="write only the *_traces.npz files (per-candidate MSE of the complete calibrations)")
    a = ap.parse_args()
    global TRACES_ONLY
    TRACES_ONLY = a.traces_only
    if not os.path.isdir(REF):
        sys.exit("make_golden.py needs the reference checkout at %s (build container only)" % REF)
    if a.tree == "all":
        for t in ("ant", "ant_wide", "olive", "olive_wide", "ant_long", "olive_long") + (() if a.traces_only else ("ant_ckpt", "olive_ckpt", "ant_mha")):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--tree", t, "--out", a.out] +
                                  (["--traces-only"] if a.traces_only else []))
        return
    import torch
    torch.set_num_threads(1)   # deterministic reductions for the recorded MSE traces
    if a.tree in ("ant_long", "olive_long"):
        gen_long(a.out, a.tree[:-5])
    elif a.tree in ("ant_ckpt", "olive_ckpt"):
        gen_ckpt(a.out, a.tree[:-5])
    elif a.tree == "ant_mha":
        gen_mha_options(a.out)
    elif a.tree == "ant":
        gen_ant(a.out)
    elif a.tree == "ant_wide":
        gen_ant_wide(a.out)
    elif a.tree == "olive_wide":
        gen_olive_wide(a.out)
    else:
        gen_olive(a.out)
    print("[make_golden] %s: quant_cuda.quant answered %d times, %d elements, C scan == independent numpy restatement on all"
          % (a.tree, SCAN_CHECKS[0], SCAN_CHECKS[1]))


if __name__ == "__main__":
    main()
