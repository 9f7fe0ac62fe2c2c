"""CPU-side product logic: grids, plan builder (host model of the device path vs the oracle
scan), C-ABI export table, module surface / model rewrite, loud failure without a GPU."""
import ctypes
import os
import re
import types

import numpy as np
import pytest

from conftest import ROOT, golden


def same_bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
    return a.shape == b.shape and bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))


def _args(**kw):
    d = dict(w_up=150, a_up=150, w_low=75, a_low=75, percent=100, search=False, no_outlier=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


# ---------------------------------------------------------------- grids
def test_product_ant_grids_bit_exact():
    from ant_quantization_amd import grids as G
    g = golden("ant_grids.npz")
    n = 0
    for k in g.files:
        inv = k.startswith("INVALID_")
        t, b, s = (k[8:] if inv else k).split("_")
        bit, signed = int(b[1:]), s == "s"
        if inv:
            with pytest.raises(Exception):
                G.ant_grid(t, bit, signed)
            continue
        mine = G.ant_grid(t, bit, signed)
        if t == "apot":
            assert np.array_equal(mine, g[k]), k     # +0/-0 order is torch.sort's unstable choice
        else:
            assert same_bits(mine, g[k]), k
        n += 1
    assert n > 100


def test_product_olive_grids_bit_exact():
    from ant_quantization_amd import grids as G
    g = golden("olive_grids.npz")
    for k in g.files:
        if k.startswith("INVALID_"):
            continue
        t, b, s = k.split("_")
        fn = {"int": G.olive_int, "flint": G.olive_flint, "outlier": G.olive_outliers}[t]
        assert same_bits(fn(int(b[1:]), s == "s"), g[k]), k


# ---------------------------------------------------------------- C ABI
def test_cabi_exports_every_declared_symbol(antq_lib):
    hdr = open(os.path.join(ROOT, "include", "antq.h")).read()
    names = set(re.findall(r"\b(antq_[a-z_0-9]+)\s*\(", hdr))
    assert {"antq_nearest", "antq_fakequant", "antq_fakequant_dynamic", "antq_plan_build", "antq_affine",
            "antq_absmax", "antq_search_sse", "antq_copy"} <= names
    L = ctypes.CDLL(antq_lib.LIB_PATH)
    for n in sorted(names):
        assert hasattr(L, n), "libantq.so does not export " + n
    ver = int(re.search(r"#define ANTQ_ABI_VERSION (\d+)", hdr).group(1))
    assert L.antq_abi_version() == ver == antq_lib.ABI_VERSION          # header, library and binding agree
    L.antq_strerror.restype = ctypes.c_char_p
    assert L.antq_strerror(-3) == b"malformed or undersized plan blob"


def test_cabi_argument_errors_without_gpu(antq_lib):
    """Entry points validate before they touch HIP: callable on a CPU-only box."""
    L = antq_lib.lib()
    assert L.antq_plan_build(None, 4, None, ctypes.c_size_t(0)) == -1
    g = np.float32([0, 1, 2, 3])
    buf = np.zeros(16, np.uint8)
    assert L.antq_plan_build(g.ctypes.data_as(ctypes.c_void_p), 4, buf.ctypes.data_as(ctypes.c_void_p),
                             ctypes.c_size_t(16)) == -3
    assert L.antq_fakequant(None, None, None, ctypes.c_size_t(4), ctypes.c_size_t(4), None, 0, ctypes.c_float(1),
                            None, None, 0, 0, None) == -1
    assert L.antq_fakequant(None, None, None, ctypes.c_size_t(0), ctypes.c_size_t(4), None, 0, ctypes.c_float(1),
                            None, None, 0, 0, None) == 0       # empty tensor: nothing to do
    assert L.antq_nearest(None, None, None, ctypes.c_size_t(0), None, 0, 0, None) == 0
    # antq_calibrate: nothing to do for an empty tensor, argument errors before any launch, workspace size arithmetic
    sz, ci = ctypes.c_size_t, ctypes.c_int
    cal = lambda rows, step, ntypes: L.antq_calibrate(None, sz(rows), sz(64), ci(1), ci(0), ci(1), None, ci(75), ci(150), ci(step),
                                                      ci(ntypes), None, None, None, ctypes.c_uint(0), None, None, None, None,
                                                      sz(0), None)
    assert cal(0, 1, 1) == 0 and cal(8, 1, 1) == -1 and cal(8, 0, 1) == -1 and cal(8, 1, 0) == -1
    wsb = lambda rows, per_row, lb, ub, step, nt: L.antq_calibrate_workspace_bytes(sz(rows), ci(per_row), ci(lb), ci(ub), ci(step), ci(nt))
    assert wsb(8, 1, 75, 150, 1, 0) == 0 and wsb(8, 1, 75, 150, 0, 2) == 0
    base = L.antq_search_workspace_bytes()
    assert wsb(8, 1, 100, 100, 1, 2) > base                       # an empty candidate range still needs the fixed parts
    assert wsb(4096, 1, 75, 150, 1, 3) - wsb(4096, 0, 75, 150, 1, 3) >= 3 * 75 * 4095 * 8      # sse: [types][cands][rows] doubles
    # ANTQ_FLAG_UNORDERED (4) with an index output: refused before anything is launched (made-up non-null addresses)
    vp = ctypes.c_void_p
    plan = antq_lib.Plan(np.float32([-1, 0, 1, 2]))
    assert L.antq_fakequant(vp(0x1000), vp(0x2000), vp(0x3000), sz(4), sz(4), vp(0x4000), 1, ctypes.c_float(2), vp(plan.host_addr),
                            vp(0x5000), ctypes.c_uint(4), 0, None) == -1
    # antq_calibrate_batch: empty batch, null job list, the workspace is the largest job's
    assert L.antq_calibrate_batch(None, ci(0), ci(0), ctypes.c_uint(0), None, sz(0), None) == 0
    assert L.antq_calibrate_batch(None, ci(3), ci(0), ctypes.c_uint(0), None, sz(0), None) == -1
    jobs = (antq_lib._CalibJob * 2)()
    for i, rows in enumerate((8, 4096)):
        jobs[i].rows, jobs[i].row_len, jobs[i].alpha_per_row = rows, 64, 1
        jobs[i].lb, jobs[i].ub, jobs[i].step, jobs[i].ntypes = 75, 150, 1, 3
    assert L.antq_calibrate_batch_workspace_bytes(jobs, ci(2)) == wsb(4096, 1, 75, 150, 1, 3)
    jobs[0].ntypes = 0
    assert L.antq_calibrate_batch_workspace_bytes(jobs, ci(2)) == 0


# ---------------------------------------------------------------- plans
def _all_grids():
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    out = {k: G[k] for k in G.files if not k.startswith("INVALID")}
    for t in ("int", "flint"):
        for b in range(3, 9):
            for s in "su":
                k = "%s_b%d_%s" % (t, b, s)
                ko = "outlier_b%d_%s" % (b, s)
                if k in O.files and ko in O.files:
                    out["olive_" + k] = np.concatenate([O[k], O[ko]])
                    out["olive_noout_" + k] = O[k]
    return out


def test_plan_host_model_equals_scan_on_every_grid(antq_lib, oracle):
    rng = np.random.default_rng(0)
    table = 0
    for k, g in _all_grids().items():
        if g.size > antq_lib.MAX_GRID:
            continue
        plan = antq_lib.Plan(g)
        table += plan.is_table
        bits = rng.integers(0, 2 ** 32, 60000, dtype=np.uint64).astype(np.uint32)
        with np.errstate(all="ignore"):
            scale = np.float32(np.nanmax(np.abs(g[np.isfinite(g)])) / 2)
            d = np.concatenate([bits.view(np.float32), rng.standard_normal(60000).astype(np.float32) * scale])
        q, idx = plan.eval_host(d)
        z, j = oracle.nearest(d, g)
        assert same_bits(q, z), k
        assert np.array_equal(idx.astype(np.int32), j), k
    assert table > 140      # nearly every grid gets the fast table path


def test_plan_exhaustive_over_all_finite_floats_headline_grid(antq_lib, oracle):
    """Every float32 in [-16, 16] (the whole reachable range of d = x/scale for |x| <= 1.6 alpha),
    i.e. ~2.2e9 bit patterns, in strided passes; plus all 2^16 bf16 patterns."""
    g = golden("ant_grids.npz")["flint_b4_s"]
    plan = antq_lib.Plan(g)
    assert plan.is_table
    hi = np.float32(16.0).view(np.uint32)
    step = 97            # co-prime stride: 11 M points per sign, adjacent-float coverage near thresholds below
    for sign in (0, 0x80000000):
        u = (np.arange(0, int(hi) + 1, step, dtype=np.uint64).astype(np.uint32) | np.uint32(sign))
        d = u.view(np.float32)
        q, idx = plan.eval_host(d)
        z, j = oracle.nearest(d, g)
        assert same_bits(q, z) and np.array_equal(idx.astype(np.int32), j)
    # a dense window of +-4096 floats around every decision threshold
    gs = np.unique(g)
    mids = ((gs[:-1].astype(np.float64) + gs[1:]) / 2).astype(np.float32)
    for m in mids:
        c = int(np.float32(abs(m)).view(np.uint32))
        u = np.arange(max(c - 4096, 0), c + 4096, dtype=np.uint64).astype(np.uint32)
        if m < 0:
            u = u | np.uint32(0x80000000)
        d = u.view(np.float32)
        q, idx = plan.eval_host(d)
        z, j = oracle.nearest(d, g)
        assert same_bits(q, z) and np.array_equal(idx.astype(np.int32), j)
    allbf = (np.arange(65536, dtype=np.uint32) << 16).view(np.float32)
    q, idx = plan.eval_host(allbf)
    z, j = oracle.nearest(allbf, g)
    assert same_bits(q, z) and np.array_equal(idx.astype(np.int32), j)


def test_plan_falls_back_to_scan_for_hostile_grids(antq_lib, oracle):
    rng = np.random.default_rng(1)
    hostile = [np.float32([3, 1, 2, 1, 3, -7]),                 # unsorted, duplicates, no zero
               np.float32([1.0, 1.0000001, 5.0]),               # near-duplicate entries (plateau)
               np.float32([0.0, np.inf, 1.0]), np.float32([np.nan, 0.0]),
               np.float32([5.0]), np.float32([-1.0, 1.0]),
               rng.standard_normal(300).astype(np.float32) * 1e4]
    d = np.concatenate([rng.standard_normal(5000).astype(np.float32) * 10,
                        rng.integers(0, 2 ** 32, 5000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    for g in hostile:
        plan = antq_lib.Plan(g)
        q, idx = plan.eval_host(d)
        with np.errstate(all="ignore"):
            z, j = oracle.nearest(d, g)
        assert same_bits(q, z) and np.array_equal(idx.astype(np.int32), j)


# ---------------------------------------------------------------- no CPU fallback
def test_loud_failure_on_cpu_tensors(antq_lib):
    import torch
    x = torch.randn(8, 64)
    plan = antq_lib.plan_for(golden("ant_grids.npz")["flint_b4_s"])
    with pytest.raises(antq_lib.AntqError, match="HIP device"):
        antq_lib.fakequant(x, torch.ones(8), plan, 10.0, 8, 64, True)
    with pytest.raises(antq_lib.AntqError, match="HIP device"):
        antq_lib.nearest(x.view(-1), torch.ones(4))
    from ant_quantization_amd.ant import quant_modules as qm
    q = qm.TensorQuantizer(mode="flint", bit=4, is_signed=True, is_enable=True, args=_args())
    with pytest.raises(antq_lib.AntqError):
        q(x)
    from ant_quantization_amd import quant_cuda
    with pytest.raises(antq_lib.AntqError):
        quant_cuda.quant(x.view(-1), torch.ones(4))


def test_loud_failure_when_library_missing(monkeypatch, antq_lib):
    monkeypatch.setattr(antq_lib, "_lib", None)
    monkeypatch.setattr(antq_lib, "LIB_PATH", "/nonexistent/libantq.so")
    with pytest.raises(antq_lib.AntqError, match="no CPU fallback"):
        antq_lib.lib()


# ---------------------------------------------------------------- module surface
@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_module_surface_and_state_dict_keys(tree):
    import importlib
    import torch
    import torch.nn as nn
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    qm = importlib.import_module("ant_quantization_amd.%s.quant_modules" % tree)
    for name in ("quantize_model", "set_first_last_layer", "set_8_bit_layer_n", "set_8_bit_layer_l",
                 "load_ant_state_dict"):
        assert hasattr(qmod, name)
    for name in ("set_quantizer", "enable_quantization", "disable_quantization", "disable_input_quantization",
                 "get_ckpt_path", "get_ckpt_filename", "set_util_logging", "get_model", "quant_args", "logging"):
        assert hasattr(qutil, name)
    assert qm.QuantConv2d is qm.Conv2dQuantizer and qm.QuantLinear is qm.LinearQuantizer

    args = _args(mode="ant-int-flint", wbit=4, abit=4)
    qutil.set_quantizer(args)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.ReLU())
            self.blocks = nn.ModuleList([nn.Linear(8, 8), nn.Linear(8, 4)])
            self.lm_head = nn.Linear(4, 2)

    m = qmod.quantize_model(Net())
    assert type(m.features[0]) is qm.Conv2dQuantizer
    assert type(m.blocks) is nn.Sequential and type(m.blocks[1]) is qm.LinearQuantizer   # ModuleList -> Sequential
    if tree == "olive":
        assert type(m.lm_head) is nn.Linear              # OliVe never descends into lm_head
    else:
        assert type(m.lm_head) is qm.LinearQuantizer
    keys = set(m.state_dict().keys())
    base = {"features.0.weight", "features.0.quant_weight.alpha", "features.0.quant_weight.bit",
            "features.0.quant_weight.has_inited_quant_para", "features.0.quant_weight.quant_grid",
            "features.0.quant_input.alpha", "blocks.0.weight", "blocks.0.bias", "blocks.1.quant_input.quant_grid"}
    assert base <= keys
    assert ("features.0.quant_weight.outliers" in keys) == (tree == "olive")
    assert m.features[0].bias is None and m.features[0].quant_weight.alpha.shape == (8, 1)
    # enable / disable walkers name the quantisers like the reference
    qutil.enable_quantization(m)
    assert m.blocks[0].quant_weight.name == "blocks.0.quant_weight" and m.blocks[0].quant_weight.is_enable
    qutil.disable_quantization(m)
    assert not m.blocks[0].quant_input.is_enable
    x = torch.randn(2, 3, 5, 5)
    assert m.features(x).shape == (2, 8, 3, 3)          # disabled quantisers are pass-through, CPU ok
    qutil.disable_input_quantization(m)
    assert not m.blocks[0].quant_input.is_enable_activation
    # re-arm policy
    qmod.set_8_bit_layer_l(m, "0")
    assert int(m.features[0].quant_weight.bit) == 8 and int(m.blocks[0].quant_weight.bit) == 4
    # load_ant_state_dict pre-sizes quant_grid so that a strict load works (8-bit layer -> 256 entries)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["features.0.quant_weight.quant_grid"] = torch.arange(256.0)
    qmod.load_ant_state_dict(m, sd)
    m.load_state_dict(sd, strict=True)
    assert m.features[0].quant_weight.quant_grid.numel() == 256


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_reference_written_checkpoint_strict_loads_on_the_host(tree):
    """The wire format of N2 without a GPU: the state dict the REFERENCE wrote (tests/golden/*_ckpt.npz, from its own
    quantize_model + calibration, keys with the DDP 'module.' prefix) strict-loads into this package's rewrite of the
    same network after load_ant_state_dict -- same key set, same shapes (OliVe: the 14-entry `outliers` and 15-entry
    `quant_grid` of a calibrated 4-bit layer, a Conv1D layer's per-channel alpha; ANT: a 256-entry grid after
    set_8_bit_layer_l), and nothing is left out.  (The forward needs the GPU: tests/test_gpu_parity.py.)"""
    import importlib
    import torch
    import torch.nn as nn
    if tree == "olive":
        pytest.importorskip("transformers")
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    fx = golden("%s_ckpt.npz" % tree)
    if tree == "ant":
        qutil.set_quantizer(_args(mode="ant-int-pot-flint", wbit=4, abit=4))
        net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Flatten(), nn.Linear(512, 32), nn.ReLU(),
                            nn.Linear(32, 10))
    else:
        from transformers import pytorch_utils
        qutil.set_quantizer(_args(mode="ant-int-flint", wbit=4, abit=4, w_up=250, a_up=250))
        net = nn.Sequential(nn.Linear(64, 128), nn.GELU(), nn.Linear(128, 64), pytorch_utils.Conv1D(32, 64))
    for prefixes in (["a__"], ["a__", "b__"]) if tree == "ant" else (["a__"],):
        check = {}
        for pre in prefixes:
            for k in fx.files:
                if k.startswith(pre + "sd__"):
                    assert k[len(pre) + 4:].startswith("module.")
                    check[k[len(pre) + 4 + 7:]] = torch.from_numpy(np.array(fx[k]))
        model = qmod.quantize_model(net)
        assert set(model.state_dict().keys()) == set(check.keys())
        qmod.load_ant_state_dict(model, check)
        res = model.load_state_dict(check, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        for k, v in model.state_dict().items():
            assert v.shape == check[k].shape and torch.equal(v, check[k]), k
        assert all(float(v) == 1.0 for k, v in check.items() if k.endswith("has_inited_quant_para"))


def test_quant_affine_helpers_match_reference_formulas():
    import torch
    from ant_quantization_amd.ant import quant_affine as qa
    x = torch.randn(4, 6)
    mn, mx = x.min(1).values, x.max(1).values
    scale, zp = qa.asymmetric_linear_quantization_params(4, mn, mx)
    assert torch.equal(scale, (mx - mn).clamp(min=1e-8).reciprocal() * 15)
    assert torch.equal(zp, (scale * mn).round() + 8)
    q = qa.linear_quantize(x, scale, zp)
    assert torch.equal(q, scale.view(-1, 1) * x - zp.view(-1, 1))
    assert torch.equal(qa.linear_dequantize(q.round(), scale, zp), (q.round() + zp.view(-1, 1)) / scale.view(-1, 1))


def test_quantize_model_on_hf_structures():
    """HF BERT / GPT-2 skeletons (random init, tiny): every Linear / Conv1D is wrapped exactly once, aliases
    such as `base_model` (a property returning a child or the module itself) neither recurse nor duplicate."""
    import torch
    transformers = pytest.importorskip("transformers")
    from transformers import BertConfig, BertForSequenceClassification, BertModel, GPT2Config, GPT2LMHeadModel
    from ant_quantization_amd.ant import quant_model as aqm, quant_modules as aq, quant_utils as aqu
    from ant_quantization_amd.olive import quant_model as oqm, quant_modules as oq, quant_utils as oqu
    args = _args(mode="ant-int-flint", wbit=4, abit=4)
    aqu.set_quantizer(args)
    oqu.set_quantizer(args)
    cfg = BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128, vocab_size=100)
    q = aqm.quantize_model(BertModel(cfg))
    assert sum(isinstance(m, aq.LinearQuantizer) for m in q.modules()) == 13
    assert sum(type(m) is torch.nn.Linear for m in q.modules()) == 0
    q = aqm.quantize_model(BertForSequenceClassification(cfg))
    assert sum(isinstance(m, aq.LinearQuantizer) for m in q.modules()) == 14
    assert not any(k.startswith("base_model.") for k in q.state_dict())
    g = GPT2LMHeadModel(GPT2Config(n_embd=64, n_layer=2, n_head=4, vocab_size=100, n_positions=32, bos_token_id=0, eos_token_id=0))
    q = oqm.quantize_model(g)
    assert sum(isinstance(m, oq.Conv1dQuantizer) for m in q.modules()) == 8
    assert type(q.lm_head) is torch.nn.Linear                 # OliVe never quantises lm_head
    oqu.disable_quantization(q)
    assert q(torch.randint(0, 100, (2, 8))).logits.shape == (2, 8, 100)


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_host_mirror_of_bit_and_inited_buffers(tree, monkeypatch):
    """`bit` / `has_inited_quant_para` are mirrored on the host (no device read per forward) but any edit of the
    buffers -- in place, by rebinding `.data` as set_8_bit_layer_* does (AQ/quant_model.py:83), through
    load_state_dict -- is noticed, and read back exactly once."""
    import importlib
    import torch
    qm = importlib.import_module("ant_quantization_amd.%s.quant_modules" % tree)
    q = qm.TensorQuantizer(mode="flint", bit=4, is_signed=True, is_enable=True, args=_args())
    reads = []
    real_item = torch.Tensor.item
    monkeypatch.setattr(torch.Tensor, "item", lambda self: (reads.append(1), real_item(self))[1])
    assert q._bits() == 4 and q._hm_get("has_inited_quant_para") == 0 and not reads      # known from construction
    q.double()                                                   # _apply re-keys, values survive: still no read
    assert q._bits() == 4 and not reads and q._hm_fresh()
    q.bit.data = torch.tensor(8)                                  # the reference's way of switching a layer to 8 bit
    assert not q._hm_fresh() and q._bits() == 8 and len(reads) == 1
    assert q._bits() == 8 and len(reads) == 1                     # ... read back once
    q.bit.fill_(6)
    assert q._bits() == 6 and len(reads) == 2
    sd = {k: v.clone() for k, v in q.state_dict().items()}
    sd["bit"] = torch.tensor(5)
    sd["has_inited_quant_para"] = torch.tensor(1.0)
    q.load_state_dict(sd)
    assert q._bits() == 5 and q._hm_get("has_inited_quant_para") == 1 and len(reads) == 4
    q._hm_known("bit", 3)                                         # what the host writes itself needs no read
    assert q._bits() == 3 and len(reads) == 4
    # the one edit the keys cannot see: through `.data`, same address, same version -- rearm() is the documented remedy
    q.bit.fill_(4)
    assert q._bits() == 4
    n0, v0, p0 = len(reads), q.bit._version, q.bit.data_ptr()
    q.bit.data.fill_(7)
    assert (q.bit._version, q.bit.data_ptr()) == (v0, p0) and q._bits() == 4 and len(reads) == n0      # unseen ...
    q.rearm()
    assert q._bits() == 7 and len(reads) == n0 + 1                                                      # ... until re-armed


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_calibration_pass_host_logic(tree):
    """The host side of the calibration pass, no kernel involved: the search memo (same tensor object, address, version and
    key only; never a Parameter; a handful of entries, dead tensors dropped), `mse` formed when read, and what a weight
    quantiser offers the batched pre-calibration (candidate types in the reference's order, duplicates once, float1-4 and
    empty windows declined, 8-bit layers = 'int' from 95)."""
    import importlib
    import torch
    from ant_quantization_amd import core
    qm = importlib.import_module("ant_quantization_amd.%s.quant_modules" % tree)
    memo = core.SearchMemo(keep=3)
    a, b = torch.randn(4, 8), torch.randn(4, 8)
    assert memo.get(a, ("k",)) is None
    memo.put(a, ("k",), "A")
    assert memo.get(a, ("k",)) == "A" and memo.hits == 1
    assert memo.get(a, ("other",)) is None and memo.get(b, ("k",)) is None and memo.get(a.clone(), ("k",)) is None
    a.add_(1.0)                                                   # version counter moved
    assert memo.get(a, ("k",)) is None
    memo.put(torch.nn.Parameter(b), ("k",), "P")                  # Parameters are never kept
    assert len(memo.entries) == 1
    for i in range(5):
        memo.put(torch.randn(2), ("t", i), i)                     # temporaries die: their entries go with the next put
    assert len(memo.entries) <= 3
    memo.enabled = False
    memo.put(b, ("k",), "B")
    assert memo.get(b, ("k",)) is None
    memo.enabled = True
    with torch.inference_mode():
        c = torch.randn(3)
    memo.put(c, ("k",), "C")                                      # no version counter: not kept
    assert memo.get(c, ("k",)) is None

    q = qm.TensorQuantizer(mode="flint", bit=4, is_signed=True, is_enable=True, args=_args())
    assert float(q.mse) == 0.0
    best = torch.tensor([1.0, 3.0])
    q._mse_later(best, 2)
    m = q.mse
    assert float(m) == 2.0 and q.mse is m                         # formed once, on the first read
    q.mse = torch.tensor(5.0)
    assert float(q.mse) == 5.0

    w = torch.randn(8, 16)
    spec = q._calib_spec(w)
    assert spec["modes"] == ["flint"] and len(spec["grids"]) == 1 and spec["step"] == (1 if tree == "ant" else 2)
    assert (spec["lb"], spec["ub"]) == (75, 150)
    qa = qm.TensorQuantizer(mode="ant-int-flint", bit=4, is_signed=True, is_enable=True, args=_args())
    sa = qa._calib_spec(w)
    assert sa["modes"] == ["int", "flint"] and len(sa["grids"]) == 2 and sa["ovp"] == (tree == "olive")
    assert sa["stat"] == ("absmax" if tree == "ant" else "3sigma")
    q8 = qm.TensorQuantizer(mode="ant-int-flint", bit=8, is_signed=True, is_enable=True, args=_args())
    s8 = q8._calib_spec(w)
    assert s8["modes"] == ["int"] and (s8["lb"] == 95 if tree == "ant" else True)
    # an input quantiser (per tensor, unsigned until a negative value was seen): what its device-side type pick searches
    qi = qm.TensorQuantizer(mode="ant-int-flint", bit=4, is_signed=False, is_enable=True, is_input=True, args=_args(a_low=60, a_up=120))
    si = qi._calib_spec(w)
    assert si["modes"] == ["int", "flint"] and (si["lb"], si["ub"]) == (60, 120) and all(float(g.min()) >= 0 for g in si["normals" if tree == "olive" else "grids"])
    qi.is_enable_activation = False
    assert qi._calib_spec(w) is None
    assert qm.TensorQuantizer(mode="flint", bit=4, is_signed=True, is_enable=True, args=_args(w_low=150, w_up=150))._calib_spec(w) is None
    assert qm.TensorQuantizer(mode="flint", bit=4, is_signed=True, is_enable=False, args=_args())._calib_spec(w) is None
    if tree == "ant":
        assert qm.TensorQuantizer(mode="ant-int-float2", bit=4, is_signed=True, is_enable=True, args=_args())._calib_spec(w) is None
        qd = qm.TensorQuantizer(mode="ant-int-pot-float-flint", bit=4, is_signed=True, is_enable=True, args=_args())
        # _TYPE_ORDER, as search_adaptive_numeric_type walks it; the 4-bit float codebook IS the pot codebook: searched once,
        # and the first of equals (pot) is what np.argsort(mse)[0] would name
        assert qd._calib_spec(w)["modes"] == ["int", "flint", "pot"] and len(qd._calib_spec(w)["grids"]) == 3


def test_dropin_directories_import_the_reference_way(tmp_path):
    """`import quant_cuda` (AQ/quant_modules.py:7) and the harnesses' `sys.path.append("../antquant"); from quant_model
    import *; from quant_utils import *` (ImageNet/main.py:14-16, llm/run_clm.py:56-59) from a clean interpreter whose
    sys.path only gains the drop-in directory: the imports succeed without a GPU and export the names the harnesses use."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    for d in ("dropin", ""):
        code = "import sys; sys.path.insert(0, %r); import quant_cuda; assert callable(quant_cuda.quant)" % \
               os.path.join(root, "ant_quantization_amd", d).rstrip("/")
        subprocess.check_call([sys.executable, "-c", code], cwd=str(tmp_path), env=env)
    names = ("set_quantizer quantize_model set_first_last_layer enable_quantization disable_quantization "
             "disable_input_quantization set_8_bit_layer_n set_8_bit_layer_l load_ant_state_dict get_ckpt_path "
             "get_ckpt_filename set_util_logging get_model logging").split()
    for tree in ("ant", "olive"):
        code = ("import sys; sys.path.append(%r)\n"
                "from quant_model import *\nfrom quant_utils import *\nimport quant_modules\n"
                "missing = [n for n in %r if n not in globals()]\nassert not missing, missing\n"
                "assert quant_modules.QuantLinear is quant_modules.LinearQuantizer\n"
                % (os.path.join(root, "ant_quantization_amd", "dropin", tree), names))
        subprocess.check_call([sys.executable, "-c", code], cwd=str(tmp_path), env=env)


def test_approximate_quotient_path_host_model_equals_oracle(antq_lib, oracle):
    """quant_vec_a (small groups / big tables): bucket and decision from x * rcp(s) with a 2^-20 margin test, exact
    redo inside the margin.  Its host model (antq_plan_eval_host_a) against the oracle's division + scan + STE sequence
    on every codebook that allows the path: inputs packed around every decision threshold moved into the x domain
    (+-64 floats), every bf16 value, random draws; many scales; the modelled reciprocal moved by -1 / 0 / +1 ulp (the
    device's v_rcp_f32 is only specified to 1 ulp: the result may not depend on it)."""
    L = antq_lib.lib()
    L.antq_plan_eval_host_a.restype = ctypes.c_int
    rng = np.random.default_rng(11)
    n_plans = n_slow = n_total = 0
    allbf = (np.arange(65536, dtype=np.uint32) << 16).view(np.float32)
    for k, g in _all_grids().items():
        if g.size > antq_lib.MAX_GRID:
            continue
        plan = antq_lib.Plan(g)
        hdr = plan.host[:96].view(np.uint32)
        if not plan.is_table or hdr[22] == 0:              # PlanHeader::adom (word 22)
            continue
        n_plans += 1
        gmax = float(np.max(g))
        gs = np.unique(g)
        mids = ((gs[:-1].astype(np.float64) + gs[1:]) / 2)
        for alpha in np.concatenate([np.float32([1.0, 0.07, 10.0, 3.3e-5, 4.1e6]),
                                     np.exp(rng.uniform(-12, 12, 3)).astype(np.float32)]):
            s = np.float32(alpha) / np.float32(gmax)
            pts = [allbf[np.isfinite(allbf)][::7], (rng.standard_normal(4000) * float(s) * gmax / 2).astype(np.float32)]
            for m in mids:
                c = np.float32(m * float(s))
                u = int(np.abs(c).view(np.uint32))
                w = np.arange(max(u - 64, 0), u + 64, dtype=np.uint64).astype(np.uint32)
                if c < 0:
                    w = w | np.uint32(0x80000000)
                pts.append(w.view(np.float32))
            x = np.ascontiguousarray(np.concatenate(pts), dtype=np.float32)
            with np.errstate(all="ignore"):
                ref, ridx = oracle.forward(x.reshape(1, -1), np.float32([alpha]), g, gmax, False)
            for ulps in (-1, 0, 1):
                out = np.empty_like(x)
                idx = np.empty(x.size, np.int16)
                slow = np.empty(x.size, np.uint8)
                rc = L.antq_plan_eval_host_a(plan.host_ptr(), x.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(x.size),
                                             ctypes.c_float(alpha), ctypes.c_float(gmax), ctypes.c_int(ulps),
                                             out.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p),
                                             slow.ctypes.data_as(ctypes.c_void_p))
                assert rc == 0
                assert same_bits(out, ref.reshape(-1)), (k, alpha, ulps)
                assert np.array_equal(idx.astype(np.int32), ridx.reshape(-1)), (k, alpha, ulps)
                n_slow += int(slow[-4000 - 0:].sum()) if False else 0
            n_total += x.size
    assert n_plans > 120, n_plans


def _half_bits_equal(got16, ref_f32, dtype, orc):
    """got16: uint16 patterns; ref_f32: the oracle's fp32 outputs -> the same rounding to 16 bits; equal bits or both NaN."""
    if dtype == 1:
        ref16 = orc.f32_to_bf16(ref_f32)
        gf, rf = orc.bf16_to_f32(got16), orc.bf16_to_f32(ref16)
    else:
        with np.errstate(all="ignore"):
            ref16 = np.asarray(ref_f32, dtype=np.float32).astype(np.float16).view(np.uint16)
        gf, rf = got16.view(np.float16).astype(np.float32), ref16.view(np.float16).astype(np.float32)
    bad = ~((got16 == ref16) | (np.isnan(gf) & np.isnan(rf)))
    return bad


def test_16bit_domain_row_path_host_model_equals_oracle_on_every_pattern(antq_lib, oracle):
    """K1h (csrc/antq_k_hrow.h): bf16 / f16 rows quantised on their bit patterns through a per-row slot table.  Its host
    model (antq_plan_eval_host_h: the same table construction, sentinel slot, far-clipped arithmetic and literal sequence
    as the kernel) against the oracle -- division, scan, pair rule, (q - d) + d, * s in fp32, then the rounding to 16 bits --
    on EVERY one of the 65 536 input patterns (in order and shuffled: other pairs), for every codebook that allows the path,
    both dtypes, with and without the pair rule, at scales from 1e-9 to 1e9 plus zero / negative / NaN / Inf alphas."""
    L = antq_lib.lib()
    L.antq_plan_eval_host_h.restype = ctypes.c_int
    rng = np.random.default_rng(23)
    allpat = np.arange(65536, dtype=np.uint16)
    shuf = rng.permutation(allpat)
    n_plans = n_table = n_far = n_rows = 0
    for k, g in _all_grids().items():
        if g.size > antq_lib.MAX_GRID:
            continue
        plan = antq_lib.Plan(g)
        hdr = plan.host[:128].view(np.uint32)
        hdom = int(hdr[24]) if plan.is_table else 0                    # PlanHeader::hdom (word 24)
        if k in ("flint_b4_s", "int_b4_s", "olive_flint_b4_s", "olive_int_b4_s", "flint_b4_u", "olive_flint_b4_u"):
            assert hdom == 3, (k, hdom)                                # the codebooks the headline configs use
        if not hdom:
            continue
        n_plans += 1
        olive = k.startswith("olive_") and not k.startswith("olive_noout_")
        gmax = float(np.max(g[np.abs(g) <= 32])) if olive else float(np.max(g))
        alphas = np.concatenate([np.float32([1.0, 0.06, 0.0, -0.05, np.nan, np.inf, 1e-30, 1e30]),
                                 np.exp(rng.uniform(-20, 20, 4)).astype(np.float32)])
        for dtype in (1, 2):
            if not (hdom >> (dtype - 1)) & 1:
                continue
            for pats in (allpat, shuf):
                xf = oracle.bf16_to_f32(pats) if dtype == 1 else pats.view(np.float16).astype(np.float32)
                for alpha in alphas:
                    for ovp in ((False, True) if olive else (False,)):
                        with np.errstate(all="ignore"):
                            ref, _ = oracle.forward(xf.reshape(1, -1), np.float32([alpha]), g, gmax, ovp)
                        out = np.empty(65536, np.uint16)
                        path = np.empty(65536, np.uint8)
                        rc = L.antq_plan_eval_host_h(plan.host_ptr(), pats.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(65536),
                                                     ctypes.c_float(alpha), ctypes.c_float(gmax), ctypes.c_int(dtype),
                                                     ctypes.c_uint(1 if ovp else 0), out.ctypes.data_as(ctypes.c_void_p),
                                                     path.ctypes.data_as(ctypes.c_void_p))
                        assert rc == 0
                        bad = _half_bits_equal(out, ref.reshape(-1), dtype, oracle)
                        assert not bad.any(), (k, dtype, float(alpha), ovp, int(bad.sum()), pats[bad][:4], out[bad][:4], path[bad][:4])
                        n_table += int((path == 0).sum())
                        n_far += int((path == 1).sum())
                        n_rows += 1
    assert n_plans >= 12 and n_rows > 500
    # (most of the 65 536 patterns lie beyond any one row's limit, and whole 8-element vectors are redone around them)
    assert n_table > 0.1 * 65536 * n_rows and n_far > 0.01 * 65536 * n_rows    # table path and far-clipped arithmetic both exercised


def test_batch_descriptor_builder_host_logic(antq_lib):
    """antq_batch_build is pure host code: every job of a batch gets its blocks exactly once, inside its family's map
    region, row jobs have tasks that cover their rows (per-tensor jobs are ONE row), mixed static batches collapse to the
    all-in-one launch, dynamic batches refuse what cannot live in registers -- checked here with made-up device addresses."""
    from ant_quantization_amd import grids
    L = antq_lib.lib()
    flint = antq_lib.Plan(grids.ant_flint(4, True))
    int8 = antq_lib.Plan(grids.ant_int(8, True))
    scan = antq_lib.Plan(np.float32([3, 1, 2, 1, 3, -7]))
    pol = antq_lib.Plan(np.concatenate([grids.olive_flint(4, True), grids.olive_outliers(4, True)]))

    def build(jobs, dtype=0, flags=0):
        arr = (antq_lib._Job * len(jobs))()
        for k, (r, c, per_row, plan) in enumerate(jobs):
            arr[k] = antq_lib._Job(0x10000000 + k * 0x4000000, 0x50000000 + k * 0x4000000, 0x30000000 + 4096 * k, r, c,
                                   1 if per_row else 0, 10.0, plan.host_addr, 0x40000000)
        cap = L.antq_batch_capacity(arr, len(jobs), dtype)
        host = np.zeros(cap, dtype=np.uint8)
        n = L.antq_batch_build(arr, len(jobs), dtype, flags, host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap))
        if n <= 0:
            return n, None
        assert n <= cap
        h = host[:80].view(np.uint32)
        total, fam, mixed, waves = int(h[7]), [int(v) for v in h[8:17]], int(h[17]) & 1, (int(h[17]) >> 8) & 7
        assert waves in (1, 4)          # wavefronts per workgroup the row-table launch will use (bits 8.. of the last word)
        assert sum(fam) == total and int(h[6]) == n == int(h[5]) + 4 * total
        bmap = host[int(h[5]):n].view(np.uint32)
        descs = []
        for k in range(len(jobs)):
            d = host[80 + 192 * k:80 + 192 * (k + 1)]
            w = d[40:72].view(np.uint32)          # total_tasks vpr tpr vshift first_block kind per_row gmax
            descs.append(dict(n_vec=int(d[32:40].view(np.uint64)[0]), total_tasks=int(w[0]), vpr=int(w[1]), tpr=int(w[2]),
                              first_block=int(w[4]), kind=int(w[5]), u=int(d[148:152].view(np.uint32)[0])))
        assert sorted(set(bmap.tolist())) == list(range(len(jobs)))              # every job owns blocks
        offs = np.cumsum([0] + fam)
        for k, d in enumerate(descs):
            pos = np.flatnonzero(bmap == k)
            assert np.array_equal(pos, np.arange(pos[0], pos[0] + pos.size))      # contiguous
            f = int(np.searchsorted(offs, pos[0], side="right") - 1)
            assert pos[0] - offs[f] == d["first_block"], (k, d)                   # first_block is relative to the family
            d["blocks"], d["family"] = int(pos.size), f
        return n, dict(descs=descs, fam=fam, mixed=mixed, lds=int(h[4]))

    # 16-bit static rows of >= 192 vectors with a 4-bit codebook that split into full tasks: the 16-bit-domain row family (5,
    # kind 13) -- rows of 576 / 192 vectors in tasks of 3 vectors per lane, powers of two in tasks of 4; rows of 144 / 128 / 288
    # vectors (short, or many idle lanes) stay with the fp32-domain row table (family 0).  A family-5 share of >= 128 MiB keeps
    # its own launch next to the other family; a small one joins the all-in-one launch as fp32-domain rows.
    for dt in (1, 2):
        n, b = build([(4096, 4608, True, flint)] * 3 + [(512, 1152, True, flint), (64, 1536, True, flint)], dtype=dt)
        assert b["mixed"] == 0 and b["fam"][5] > 0 and b["fam"][0] > 0 and sum(b["fam"]) == b["fam"][5] + b["fam"][0]
        assert [d["u"] for d in b["descs"]] == [3, 3, 3, 3, 3] and [d["kind"] for d in b["descs"]] == [13, 13, 13, 2, 13]
        for d, rows in zip(b["descs"], (4096, 4096, 4096, 512, 64)):
            assert d["total_tasks"] == rows * d["tpr"] and d["tpr"] * 64 * d["u"] >= d["vpr"] and d["blocks"] == -(-d["total_tasks"] // 4)
        n, b = build([(4096, 4096, True, flint)] * 3 + [(512, 2048, True, flint), (8, 1 << 16, False, pol)], dtype=dt)
        assert b["mixed"] == 0 and b["fam"][5] > 0 and sum(b["fam"]) == b["fam"][5] and all(d["kind"] == 13 for d in b["descs"])
        assert [d["u"] for d in b["descs"]] == [4, 4, 4, 4, 4] and b["descs"][4]["tpr"] == b["descs"][4]["total_tasks"] == 256
        n, b = build([(64, 4096, True, flint), (512, 1024, True, flint), (64, 2304, True, flint)], dtype=dt)     # small: one family-0 launch
        assert b["mixed"] == 0 and b["fam"][0] == sum(b["fam"]) and all(d["kind"] == 2 for d in b["descs"])
        n, b = build([(4096, 4096, True, flint)] * 4 + [(512, 1024, True, flint), (64, 147, True, flint), (64, 64, True, flint)], dtype=dt)
        assert b["mixed"] == 1 and b["fam"][5] > 0 and b["fam"][0] > 0 and sum(b["fam"]) == b["fam"][5] + b["fam"][0]
        assert [d["kind"] for d in b["descs"]] == [13, 13, 13, 13, 2, 3, 1]      # big family 5 apart, the rest all-in-one
    # ... with knob 9 = 0 the fp32-domain row table (family 0, kind 2) as in round 3
    L.antq_debug_set(9, 0)
    n, b = build([(4096, 4608, True, flint)] * 3 + [(512, 1152, True, flint), (64, 1536, True, flint)], dtype=1)
    assert b["mixed"] == 0 and b["fam"][0] > 0 and sum(b["fam"][1:]) == 0
    assert [d["u"] for d in b["descs"]] == [3, 3, 3, 3, 3] and all(d["kind"] == 2 for d in b["descs"])
    n, b = build([(4096, 4096, True, flint)] * 3 + [(512, 1024, True, flint)], dtype=1)
    assert b["mixed"] == 0 and b["fam"][0] > 0 and sum(b["fam"][1:]) == 0 and all(d["kind"] == 2 for d in b["descs"])
    assert [d["u"] for d in b["descs"]] == [2, 2, 2, 2]      # (round 3: one-wavefront workgroups stream best with 2 KiB each)
    L.antq_debug_set(9, 1)
    n, b = build([(4096, 4096, True, flint)] * 3 + [(512, 1024, True, flint)], dtype=0)
    assert b["mixed"] == 0 and b["fam"][1] > 0 and b["fam"][0] == 0 and all(d["kind"] == 1 for d in b["descs"])
    assert [d["n_vec"] for d in b["descs"]] == [4096 * 1024] * 3 + [512 * 256]
    assert all(d["blocks"] == -(-d["n_vec"] // (256 * d["u"])) for d in b["descs"])
    # per-tensor scale: ONE row however the caller shaped the tensor
    n, b = build([(256, 512, False, pol), (64, 147, False, pol), (7, 33, False, pol)], flags=1)
    assert b["descs"][0]["kind"] == 1 and b["descs"][0]["vpr"] == 32768 and b["descs"][0]["n_vec"] == 32768   # 2^15 vectors: a lane job
    assert b["descs"][1]["kind"] == 1 and b["descs"][1]["n_vec"] == 2352          # 9408 fp32 elements: whole vectors, a lane job
    assert b["descs"][2]["kind"] == 3 and b["mixed"] == 0 and b["fam"][1] == sum(d["blocks"] for d in b["descs"])
    # ResNet-50 per channel, fp32: every aligned row is a lane job, conv1's ragged K = 147 rides along: one family
    n, b = build([(64, 147, True, flint), (64, 64, True, flint), (256, 576, True, flint), (2048, 512, True, flint)])
    assert b["mixed"] == 0 and b["fam"][1] == sum(d["blocks"] for d in b["descs"]) and b["lds"] > 0
    assert [d["kind"] for d in b["descs"]] == [3, 1, 1, 1]
    # the same in bf16: rows of 144 vectors keep their per-row table -> long rows, short rows, ragged rows: one all-in-one launch
    n, b = build([(64, 147, True, flint), (64, 64, True, flint), (256, 1152, True, flint), (2048, 512, True, flint)], dtype=1)
    assert b["mixed"] == 1 and b["fam"][0] == sum(d["blocks"] for d in b["descs"]) and b["lds"] > 0
    assert [d["kind"] for d in b["descs"]] == [3, 1, 2, 1]
    # group-16 (all lane jobs, adom): family 1 alone; a scan plan: family 2; both together: mixed
    n, b = build([(1 << 16, 16, True, flint)] * 2, dtype=1)
    assert b["mixed"] == 0 and b["fam"][1] > 0 and b["fam"][0] == 0 and all(d["kind"] == 1 for d in b["descs"])
    n, b = build([(1 << 12, 64, True, scan)])
    assert b["fam"][2] > 0 and b["mixed"] == 0
    n, b = build([(1 << 12, 64, True, scan), (1 << 12, 64, True, flint)])
    assert b["mixed"] == 1
    # big tables (int-8: no per-row copy) through the exact-decision path: rows per wavefront, or lane jobs when the row
    # is a power of two of vectors
    n, b = build([(512, 4608, True, int8), (512, 4096, True, int8)], dtype=1)
    assert [d["kind"] for d in b["descs"]] == [0, 0] and b["fam"][1] > 0 and b["lds"] >= 255 * 16
    n, b = build([(512, 4608, True, int8), (512, 4096, True, int8)])
    assert [d["kind"] for d in b["descs"]] == [1, 1] and b["fam"][1] > 0 and b["lds"] >= 255 * 16
    # dynamic: groups (power of two, <= 64 vectors), rows in a wavefront / a workgroup / a 1024-thread workgroup
    n, b = build([(1 << 14, 16, True, flint), (4096, 512, True, flint), (512, 4096, True, flint), (64, 28672, True, flint),
                  (16, 65536, True, flint), (300, 2048, True, flint)], dtype=1, flags=2)
    # (16-bit rows of 128 .. 8192 vectors: the 16-bit-domain kernels -- the row in 1 / 4 / 16 wavefronts, families 6 / 7 / 8)
    assert [d["kind"] for d in b["descs"]] == [1, 1, 14, 16, 16, 14] and [d["family"] for d in b["descs"]] == [1, 1, 6, 8, 8, 6]
    assert [d["u"] for d in b["descs"]][2:] == [8, 4, 8, 4]
    assert b["descs"][3]["blocks"] == 64 and b["descs"][2]["blocks"] == 128 and b["descs"][5]["blocks"] == 75
    n, b = build([(100, 8192, True, flint), (100, 16384, True, flint)], dtype=2, flags=2)
    assert [d["kind"] for d in b["descs"]] == [15, 15] and [d["u"] for d in b["descs"]] == [4, 8] and b["fam"][7] == 200
    L.antq_debug_set(9, 0)             # ... and the round-3 kernels without them
    n, b = build([(1 << 14, 16, True, flint), (4096, 512, True, flint), (512, 4096, True, flint), (64, 28672, True, flint),
                  (16, 65536, True, flint), (300, 2048, True, flint)], dtype=1, flags=2)
    assert [d["kind"] for d in b["descs"]] == [1, 1, 6, 9, 10, 0] and [d["family"] for d in b["descs"]] == [1, 1, 3, 4, 4, 1]
    L.antq_debug_set(9, 1)
    for bad in ([(8, 576, True, flint)], [(8, 65544, True, flint)], [(8, 147, True, flint)], [(64, 64, False, flint)]):
        n, _ = build(bad, dtype=1, flags=2)
        assert n == -2, bad                              # ANTQ_ERR_UNSUPPORTED
    # 16-bit dynamic rows of 128 vectors: a row per wavefront in the 16-bit domain; without those kernels lane jobs whose
    # groups span 2 wavefronts (4 vectors per lane, never 2)
    n, b = build([(4096, 1024, True, flint)], dtype=1, flags=2)
    assert b["descs"][0]["kind"] == 14 and b["descs"][0]["vpr"] == 128 and b["descs"][0]["u"] == 2 and b["descs"][0]["family"] == 6
    L.antq_debug_set(9, 0)
    n, b = build([(4096, 1024, True, flint)], dtype=1, flags=2)
    assert b["descs"][0]["kind"] == 1 and b["descs"][0]["vpr"] == 128 and b["descs"][0]["u"] == 4 and b["descs"][0]["family"] == 1
    L.antq_debug_set(9, 1)
    # alpha_dev: required for a static job, optional (the scales are simply not stored) for a dynamic one
    arr = (antq_lib._Job * 1)()
    for flags, want in ((0, -1), (2, None)):
        arr[0] = antq_lib._Job(0x10000000, 0x50000000, 0, 4096, 128, 1, 10.0, flint.host_addr, 0x40000000)
        cap = L.antq_batch_capacity(arr, 1, 1)
        host = np.zeros(cap, dtype=np.uint8)
        rc = L.antq_batch_build(arr, 1, 1, flags, host.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap))
        assert (rc == want) if want is not None else rc > 0, (flags, rc)


def test_pinned_slot_pool_never_hands_out_a_slot_that_is_still_owned():
    """_mirror._PinnedSlots: a slot belongs to its taker until given back; the pool grows instead of wrapping (a model with
    more quantised layers than one chunk keeps every parked type pick intact)."""
    import torch
    from ant_quantization_amd._mirror import _PinnedSlots
    pool = _PinnedSlots(chunk=8, alloc=lambda n: torch.zeros(n))
    held = [pool.take() for _ in range(50)]
    for i, s in enumerate(held):
        s[0] = float(i)
    assert len(pool.chunks) == 7 and len({s.data_ptr() for s in held}) == 50
    assert [int(s[0]) for s in held] == list(range(50))
    for s in held[::2]:
        pool.give(s)
    again = [pool.take() for _ in range(25)]
    assert {s.data_ptr() for s in again} == {s.data_ptr() for s in held[::2]} and len(pool.chunks) == 7
    for s in again:
        s[0] = -1.0
    assert [int(s[0]) for s in held[1::2]] == list(range(1, 50, 2))
    pool.give(None)
    # (1) two threads taking at once never see the same slot or an IndexError; (2) a slot taken for an owner
    # that dies without giving it back returns by itself; giving twice is harmless
    import gc
    import threading
    pool2 = _PinnedSlots(chunk=4, alloc=lambda n: torch.zeros(n))
    got, errs = [[], []], []

    def worker(k):
        try:
            for _ in range(500):
                got[k].append(pool2.take())
        except Exception as ex:          # noqa: BLE001
            errs.append(ex)

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs and len({s.data_ptr() for s in got[0] + got[1]}) == 1000 and pool2.outstanding() == 1000
    for s in got[0] + got[1]:
        pool2.give(s)
    pool2.give(got[0][0])
    assert pool2.outstanding() == 0 and len(pool2.free) == len(pool2.chunks) * 4

    class Owner:
        pass
    o = Owner()
    s1, s2 = pool2.take(o), pool2.take(o)
    pool2.give(s1)
    assert pool2.outstanding() == 1
    del o
    gc.collect()
    assert pool2.outstanding() == 0 and len(pool2.free) == len(pool2.chunks) * 4


def test_multihead_attention_wrapper_equals_torch_attention_with_every_option():
    """ant/multihead_attention.py with its quantisers switched off is torch's nn.MultiheadAttention (self-attention): the
    options the reference forwards to its vendored torch 1.11 code -- add_bias_kv, add_zero_attn, batch_first, boolean and
    float masks, unbatched input -- all give torch's own output and attention weights (CPU: no kernel is involved)."""
    import types
    import torch
    import torch.nn as nn
    from ant_quantization_amd.ant import quant_model as qmod, quant_utils as qutil
    from ant_quantization_amd.ant.multihead_attention import MultiheadAttentionQuantizer
    qutil.set_quantizer(types.SimpleNamespace(mode="flint", wbit=4, abit=4, w_up=150, a_up=150, w_low=75, a_low=75, percent=100,
                                              search=False))
    torch.manual_seed(3)
    for kw in (dict(), dict(add_bias_kv=True), dict(add_zero_attn=True), dict(add_bias_kv=True, add_zero_attn=True, batch_first=True),
               dict(bias=False, add_zero_attn=True)):
        ma = nn.MultiheadAttention(32, 4, **kw).eval()
        wrapped = qmod.quantize_model(nn.Sequential(ma)).eval()
        assert type(wrapped[0]) is MultiheadAttentionQuantizer
        qutil.disable_quantization(wrapped)
        bf = kw.get("batch_first", False)
        x = torch.randn(2, 7, 32) if bf else torch.randn(7, 2, 32)
        kpm = torch.zeros(2, 7, dtype=torch.bool)
        kpm[1, 5:] = True
        for masks in (dict(), dict(key_padding_mask=kpm), dict(attn_mask=torch.randn(7, 7)),
                      dict(attn_mask=torch.triu(torch.ones(7, 7, dtype=torch.bool), 1), key_padding_mask=kpm),
                      dict(attn_mask=torch.randn(8, 7, 7) * 0.3)):
            for avg in (True, False):
                with torch.no_grad():
                    y0, w0 = ma(x, x, x, need_weights=True, average_attn_weights=avg, **masks)
                    y1, w1 = wrapped[0](x, x, x, need_weights=True, average_attn_weights=avg, **masks)
                torch.testing.assert_close(y1, y0, rtol=1e-5, atol=1e-6)
                torch.testing.assert_close(w1, w0, rtol=1e-5, atol=1e-6)
        xu = torch.randn(7, 32)                       # unbatched
        with torch.no_grad():
            y0, w0 = ma(xu, xu, xu)
            y1, w1 = wrapped[0](xu, xu, xu)
        torch.testing.assert_close(y1, y0, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(w1, w0, rtol=1e-5, atol=1e-6)
    with pytest.raises(NotImplementedError):
        qmod.quantize_model(nn.Sequential(nn.MultiheadAttention(32, 4, kdim=16, vdim=16)))
