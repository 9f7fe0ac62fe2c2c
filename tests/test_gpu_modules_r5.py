"""Module-level behaviour added in round 5 (GPU): the default weight bank under the reference's own train / eval loops, the
pinned-slot pool behind deferred type picks on deep models, the bank's memory gate.

Reference behaviour matched: nothing is cached between forwards (AQ:613-617, :642-646; OQ:413-416, :443-446), the BERT harness
steps with `p.data.add_(-update_with_lr)` (BERT/optimization.py:161) and evaluates under torch.no_grad() after every epoch
(BERT/run_glue.py:596-668)."""
import copy
import importlib
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _args(**kw):
    d = dict(w_up=150, a_up=150, w_low=75, a_low=75, percent=100, search=False, no_outlier=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def _trees(tree):
    return (importlib.import_module("ant_quantization_amd.%s.quant_model" % tree),
            importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree))


def _bert_adam_step(model, lr=0.05, wd=0.01):
    """The update rule of the reference's BertAdam as far as version counters are concerned (BERT/optimization.py:150-161):
    every write goes through `.data` -- no Parameter's `_version` moves.  Weight decay makes every parameter (weights AND
    alphas) move even where no gradient arrives (the OliVe tree quantises under no_grad)."""
    import torch
    for p in model.parameters():
        update = wd * p.data
        if p.grad is not None:
            update = update + p.grad.data
            p.grad = None
        v0 = p._version
        p.data.add_(-lr * update)
        assert p._version == v0


@pytest.mark.parametrize("schedule", ["default", "resident"])
@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_run_glue_epoch_loop_with_data_writing_optimiser(antq_lib, dev, tree, dtype_name, schedule, capsys):
    """calibrate -> eval -> [train steps whose optimiser writes p.data.add_() on weights and alphas -> eval] x 3, the default
    module path (AutoBank armed by enable_quantization) against the reference's schedule on the same parameters:
    bit-identical outputs after every epoch.  Default schedule: ONE bank launch per no-grad forward (the reference
    re-quantises every weight on every forward, AQ:613-617); resident (set_weights_at_rest): ONE per evaluation phase."""
    import torch
    import torch.nn as nn
    qmod, qutil = _trees(tree)
    dt = getattr(torch, dtype_name)
    qutil.set_quantizer(_args(mode="flint", wbit=4, abit=4))
    torch.manual_seed(5)
    net = nn.Sequential(nn.Linear(256, 512), nn.GELU(), nn.Linear(512, 512), nn.GELU(), nn.Linear(512, 256), nn.Linear(256, 4))
    model = qmod.quantize_model(net).to(dev).to(dt)
    capsys.readouterr()
    qutil.enable_quantization(model)
    resident = schedule == "resident"
    if resident:
        qutil.set_weights_at_rest(model, True)
    ab = model._antq_auto_bank
    x = torch.randn(64, 256, device=dev).to(dt)
    xt = torch.randn(32, 256, device=dev).to(dt)
    model.eval()
    with torch.no_grad():
        model(x)                              # calibration
        y_prev = model(x)                     # epoch-0 evaluation: the bank attaches
        assert ab.bank is not None and ab.bank.launches == 1 and ab.bank.resident == resident
        model(x)
        # resident: unchanged weights under no_grad, nothing is launched for them; default: one launch per forward
        assert ab.bank.launches == (1 if resident else 2)
    for epoch in range(3):
        bank = ab.bank
        before = bank.launches
        model.train()
        for _ in range(2):
            out = model(xt)
            out.float().pow(2).mean().backward()
            _bert_adam_step(model)
        assert ab.bank is bank and bank.launches == before        # training forwards never touch the resident copies
        model.eval()
        with torch.no_grad():
            y_bank = model(x)
            assert bank.launches == before + 1
            y_again = model(x)
            assert bank.launches == before + (1 if resident else 2) and torch.equal(y_bank, y_again)
            ref = copy.deepcopy(model)                            # no bank travels with a copy: per-layer schedule
            assert all(m.quant_weight._bank is None for m in ref.modules() if hasattr(m, "quant_weight"))
            y_ref = ref(x)
            qutil.set_weight_bank(model, False)                   # ... and the reference's schedule on the model itself
            y_ref2 = model(x)
            qutil.set_weight_bank(model, True)
            model(x)                                              # (the bank of this epoch's weights: stale after the next one)
            assert ab.bank is not None
        assert torch.equal(y_bank, y_ref) and torch.equal(y_bank, y_ref2), (tree, dtype_name, epoch)
        assert not torch.equal(y_bank, y_prev), "the optimiser moved nothing: the test would prove nothing"
        y_prev = y_bank
    capsys.readouterr()


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_plain_data_edit_between_two_no_grad_forwards_is_seen_by_default(antq_lib, dev, tree, capsys):
    """The reference caches nothing: every forward re-quantises every weight (AQ:613-617, :642-646), so `w.data.mul_()`
    between two no-grad forwards -- no train() / eval(), no training forward, no version counter moved -- changes the next
    output.  The default module path (one batched refresh per no-grad forward) does the same, bit-identical to a fresh copy
    of the model; layers called directly (not through the model's forward) see their edits too.  Only the opt-in resident
    mode (set_weights_at_rest: "nothing writes the weights behind torch's back") keeps its copies: there the same edit
    needs bank.invalidate(), while train() / eval() transitions still refresh."""
    import torch
    import torch.nn as nn
    qmod, qutil = _trees(tree)
    qutil.set_quantizer(_args(mode="int", wbit=4, abit=8))
    torch.manual_seed(6)
    model = qmod.quantize_model(nn.Sequential(nn.Linear(128, 128), nn.ReLU(), nn.Linear(128, 16))).to(dev).eval()
    capsys.readouterr()
    qutil.enable_quantization(model)
    x = torch.randn(32, 128, device=dev)
    lin = [m for m in model.modules() if hasattr(m, "quant_weight")]
    with torch.no_grad():
        model(x)
        y0 = model(x)
        bank = model._antq_auto_bank.bank
        assert bank is not None and bank.launches == 1 and not bank.resident
        v0 = lin[1].weight._version
        lin[1].weight.data.mul_(0.5)                                  # the edit nothing can see ...
        assert lin[1].weight._version == v0
        y1 = model(x)
        assert bank.launches == 2 and not torch.equal(y1, y0)         # ... is seen: this forward re-quantised
        fresh = copy.deepcopy(model)
        qutil.set_weight_bank(fresh, False)                           # the reference's per-layer schedule on a fresh copy
        assert torch.equal(y1, fresh(x))
        # layers called directly, twice, with an edit in between: the second call re-quantises (no model forward around it)
        h = torch.randn(8, 128, device=dev)
        a0 = lin[0](h)
        lin[0].weight.data.mul_(1.5)
        a1 = lin[0](h)
        assert not torch.equal(a0, a1) and torch.equal(a1, copy.deepcopy(lin[0])(h))
        # resident mode, opt-in: (1) train() / eval() around `.data` writes with no forward at all (an EMA copy, a checkpoint
        # averaged into place) refresh; (2) the bare edit is the caller's promise not to happen -- invalidate() says it did
        qutil.set_weights_at_rest(model, True)
        y2 = model(x)
        n = bank.launches
        assert model._antq_auto_bank.bank is bank and bank.resident and torch.equal(model(x), y2) and bank.launches == n
        model.train()
        lin[0].weight.data.mul_(1.5)
        model.eval()
        y3 = model(x)
        assert bank.launches == n + 1 and not torch.equal(y3, y2)
        lin[1].weight.data.mul_(0.5)
        y_stale = model(x)
        assert bank.launches == n + 1 and torch.equal(y_stale, y3)
        bank.invalidate()
        y4 = model(x)
        assert bank.launches == n + 2 and not torch.equal(y4, y3)
        qutil.set_weights_at_rest(model, False)                       # back to the default: every forward re-quantises
        assert not bank.resident and torch.equal(model(x), y4) and bank.launches == n + 3
        lin[1].weight.data.mul_(2.0)
        assert not torch.equal(model(x), y4) and bank.launches == n + 4
    capsys.readouterr()


@pytest.mark.parametrize("resident", [True, False])
@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_weights_at_rest_mode_after_a_training_step(antq_lib, dev, tree, resident, capsys):
    """set_weights_at_rest (opt-in; resident bank, or per-layer unordered launches with resident=False; bf16 model: alpha kept as a float32 copy keyed by (address, version)): a training step
    that writes weight and alpha through `.data`, then evaluation -- the copy of alpha is re-read and the first launch is
    ordered, outputs bit-identical to a fresh copy of the model without the mode."""
    import torch
    import torch.nn as nn
    qmod, qutil = _trees(tree)
    qutil.set_quantizer(_args(mode="flint", wbit=4, abit=4))
    torch.manual_seed(8)
    model = qmod.quantize_model(nn.Sequential(nn.Linear(256, 256), nn.ReLU(), nn.Linear(256, 8))).to(dev).to(torch.bfloat16)
    capsys.readouterr()
    qutil.enable_quantization(model)
    qutil.set_weights_at_rest(model, True, resident=resident)
    x = torch.randn(16, 256, device=dev).to(torch.bfloat16)
    model.eval()
    with torch.no_grad():
        model(x)
        model(x)
        y0 = model(x)
    for _ in range(2):
        model.train()
        model(x).float().pow(2).mean().backward()
        _bert_adam_step(model, lr=0.2)
        model.eval()
        with torch.no_grad():
            y1 = model(x)
            y2 = model(x)
            ref = copy.deepcopy(model)
            qutil.set_weights_at_rest(ref, False)
            assert torch.equal(y1, ref(x)) and torch.equal(y1, y2)
        assert not torch.equal(y0, y1)
        y0 = y1
    capsys.readouterr()


@pytest.mark.parametrize("tree,mode", [("ant", "ant-int-pot-flint"), ("olive", "ant-int-flint")])
def test_deferred_type_picks_on_a_260_layer_model(antq_lib, dev, tree, mode, capsys):
    """More quantised layers than any fixed ring of pinned slots would hold (two slots per layer: sign probe + parked type
    pick; GPT-2 XL / OPT have 192 linears): every pick parked until the model-level flush is still its own when it is read.
    Same modes, codebooks, alphas and outputs as with every pick read on the spot."""
    import torch
    import torch.nn as nn
    qmod, qutil = _trees(tree)
    from ant_quantization_amd import _mirror as _m0
    taken_before = _m0._slots.outstanding()
    qutil.set_quantizer(_args(mode=mode, wbit=4, abit=4))

    class Deep(nn.Module):
        def __init__(self, n=260, w=64):
            super().__init__()
            self.layers = nn.ModuleList([nn.Linear(w, w) for _ in range(n)])

        def forward(self, h):
            for i, l in enumerate(self.layers):
                z = l(h)
                # (different activation statistics per layer so that the picks differ along the depth)
                h = (torch.tanh(z) if i % 3 == 0 else torch.relu(z) if i % 3 == 1 else z * z.abs()) + 0.5 * h
                h = h / h.abs().amax().clamp_min(1e-6)
            return h

    res = []
    for defer in (True, False):
        torch.manual_seed(9)
        m = qmod.quantize_model(Deep()).to(dev).eval()
        capsys.readouterr()
        qutil.enable_quantization(m)
        m._antq_auto_bank.defer_types = defer
        x = torch.randn(96, 64, device=dev)
        with torch.no_grad():
            y = m(x)
            log = capsys.readouterr().out
            y2 = m(x)
        qs = [l.quant_input for l in m.modules() if hasattr(l, "quant_input")]
        assert len(qs) == 260 and all(q._pending is None and q._steady for q in qs)
        if defer:
            assert m._antq_auto_bank.deferred == 260
        res.append((y, y2, log, [(q.mode, q.quant_grid.clone(), q.alpha.detach().clone(), q._gmax) for q in qs]))
    (ya, ya2, la, sa), (yb, yb2, lb, sb) = res
    assert la == lb and la.count("-bit") == 520
    assert len({s[0] for s in sa}) >= 2, "every layer picked the same type: the test would not see a mixed-up slot"
    for i, (a, b) in enumerate(zip(sa, sb)):
        assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and a[3] == b[3], (tree, i, a[0], b[0])
    assert torch.equal(ya, yb) and torch.equal(ya2, yb2) and torch.equal(ya, ya2)
    from ant_quantization_amd import _mirror
    pool = _mirror._slots
    assert pool.outstanding() == taken_before, "a pinned slot was not given back"
    assert len(pool.free) + pool.outstanding() == len(pool.chunks) * pool.chunk


def test_weight_bank_memory_gate_and_failure_are_permanent(antq_lib, dev, capsys, monkeypatch):
    """The resident copies are only built when they fit comfortably (ANTQ_BANK_MEM_FRACTION of the free device memory);
    a model that does not fit -- or an allocation failure while building -- leaves the per-layer schedule in place, for
    good: no retry on every forward, nothing half-attached, the forward's result unchanged."""
    import torch
    import torch.nn as nn
    from ant_quantization_amd import weight_bank
    qmod, qutil = _trees("ant")
    qutil.set_quantizer(_args(mode="flint", wbit=4, abit=4))
    x = torch.randn(8, 128, device=dev)

    def make():
        torch.manual_seed(3)
        m = qmod.quantize_model(nn.Sequential(nn.Linear(128, 256), nn.ReLU(), nn.Linear(256, 8))).to(dev).eval()
        capsys.readouterr()
        qutil.enable_quantization(m)
        return m

    m0 = make()
    with torch.no_grad():
        m0(x)
        y_bank = m0(x)
    assert m0._antq_auto_bank.bank is not None
    # (a) the gate
    m1 = make()
    m1._antq_auto_bank.mem_fraction = 0.0
    with torch.no_grad():
        m1(x)
        y1 = m1(x)
    ab = m1._antq_auto_bank
    assert ab.bank is None and not ab.enabled and "MB free" in ab.reason
    assert all(l.quant_weight._bank is None for l in m1.modules() if hasattr(l, "quant_weight"))
    assert torch.equal(y1, y_bank)
    # (b) out of memory while the bank is being built
    m2 = make()
    with torch.no_grad():
        m2(x)
    built, state = [], {"n": 0, "armed": True}
    real = torch.empty_like

    class Counting(weight_bank.WeightBank):
        def __init__(self, model, **kw):
            built.append(1)
            super().__init__(model, **kw)

    def failing(t, *a, **k):
        if state["armed"]:
            state["n"] += 1
            if state["n"] == 2:                                       # the second layer's resident copy
                state["armed"] = False
                raise torch.cuda.OutOfMemoryError("HIP out of memory (simulated)")
        return real(t, *a, **k)

    monkeypatch.setattr(weight_bank, "WeightBank", Counting)
    monkeypatch.setattr(weight_bank.torch, "empty_like", failing)
    with torch.no_grad():
        y2 = m2(x)
    monkeypatch.setattr(weight_bank.torch, "empty_like", real)
    ab = m2._antq_auto_bank
    assert ab.bank is None and not ab.enabled and "failed" in ab.reason and built == [1] and not state["armed"]
    assert all(l.quant_weight._bank is None for l in m2.modules() if hasattr(l, "quant_weight"))
    with torch.no_grad():
        assert torch.equal(m2(x), y_bank) and torch.equal(y2, y_bank)
    assert built == [1]                                               # no retry
    monkeypatch.undo()
    # (c) out of memory during a later refresh: the bank gives up, the forward still answers
    m3 = make()
    with torch.no_grad():
        m3(x)
        m3(x)
        bank = m3._antq_auto_bank.bank
        assert bank is not None
        bank.invalidate()

        def boom():
            raise torch.cuda.OutOfMemoryError("HIP out of memory (simulated)")
        bank.refresh = bank._refresh_fast = boom
        with pytest.warns(UserWarning, match="weight bank switched off"):
            y3 = m3(x)
        assert torch.equal(y3, y_bank) and m3._antq_auto_bank.bank is None and not m3._antq_auto_bank.enabled
        assert torch.equal(m3(x), y_bank)
    capsys.readouterr()


def test_prewarm_leaves_the_global_rng_alone(antq_lib, dev):
    import torch
    from ant_quantization_amd import _lib
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    a = torch.randn(4, device=dev)
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    _lib._prewarmed.discard((dev.index, torch.float16))
    _lib.prewarm(dev, torch.float16)
    b = torch.randn(4, device=dev)
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------------------------------
# K1h (csrc/antq_k_hrow.h) along the SCALE axis
# ---------------------------------------------------------------------------------------------------------------------------
def _f32_bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def _to16(v, dtype):
    """Nearest 16-bit magnitude pattern of positive float32 values (sampling aid only: neighbours are added around it)."""
    v = np.asarray(v, dtype=np.float32)
    with np.errstate(all="ignore"):
        if dtype == 1:
            return np.minimum((_f32_bits(v).astype(np.uint64) + 0x8000) >> 16, 0x7f80).astype(np.uint32)
        return np.minimum(v, 65504.0).astype(np.float16).view(np.uint16).astype(np.uint32)


def _hrow_has_table(L, plan, alpha, gmax, dtype, ovp):
    """Does the row of this scale get a slot table (hrow_build's `fast`), by the library's own host model of the path?"""
    import ctypes
    x = np.zeros(8, np.uint16)
    out, path = np.empty(8, np.uint16), np.empty(8, np.uint8)
    rc = L.antq_plan_eval_host_h(plan.host_ptr(), x.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(8), ctypes.c_float(alpha),
                                 ctypes.c_float(gmax), ctypes.c_int(dtype), ctypes.c_uint(1 if ovp else 0),
                                 out.ctypes.data_as(ctypes.c_void_p), path.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return bool(path[0] == 0)


def _flip_scales(L, plan, gmax, dtype, ovp):
    """Every float32 alpha at which the row's table appears or disappears, to the last bit: a coarse sweep of the whole
    positive float32 range (four points per binade) and a bisection on the bit pattern between two points that disagree.
    Returns [(last alpha of one kind, first alpha of the other)]."""
    grid = np.concatenate([np.ldexp(np.float32(1.0 + j / 4.0), np.arange(-149, 128)) for j in range(4)]).astype(np.float32)
    grid = np.unique(grid[np.isfinite(grid) & (grid > 0)])
    state = [_hrow_has_table(L, plan, float(a), gmax, dtype, ovp) for a in grid]
    flips = []
    for i in range(len(grid) - 1):
        if state[i] != state[i + 1]:
            lo, hi = int(_f32_bits(grid[i])), int(_f32_bits(grid[i + 1]))
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if _hrow_has_table(L, plan, float(np.uint32(mid).view(np.float32)), gmax, dtype, ovp) == state[i]:
                    lo = mid
                else:
                    hi = mid
            flips.append((np.uint32(lo).view(np.float32), np.uint32(hi).view(np.float32)))
    return flips


def test_16bit_domain_row_kernels_along_the_scale_axis(antq_lib, oracle, dev):
    """K1h's per-row table depends on the row's scale: which slot a threshold lands in, where the sentinel slot sits, whether
    the row gets a table at all (hrow_build's `fast`, antq_k_hrow.h:92-116).  test_16bit_domain_row_kernels_on_every_pattern
    covers every PATTERN at 16 scales; this one covers the SCALE axis: 256 log-uniform alphas plus the exact float32
    neighbours (both sides, +- 1 ulp) of every alpha at which `fast` flips, each row holding the patterns that can go
    wrong at its scale -- every slot boundary of the row's key range +- 1, every threshold's pattern +- 3, the row limit
    +- 3, zeros / Inf / NaN / denormals / the largest finite values, and random patterns in random pair positions --
    against the oracle's fp32 sequence rounded to 16 bits, bit for bit: one launch per tensor (4- and 8-vector tasks, ordered and
    unordered), the batched launch, ANT and OliVe pairs, bf16 and f16."""
    import torch
    from conftest import golden
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    L = antq_lib.lib()
    import ctypes
    L.antq_plan_eval_host_h.restype = ctypes.c_int
    rng = np.random.default_rng(77)
    books = [("flint_b4_s", G["flint_b4_s"], None, False), ("int_b4_s", G["int_b4_s"], None, False),
             ("flint_b4_u", G["flint_b4_u"], None, False), ("pot_b4_s", G["pot_b4_s"], None, False),
             ("float_b5_s", G["float_b5_s"], None, False),
             ("olive_flint_b4_s", np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]]), float(O["flint_b4_s"].max()), True),
             ("olive_int_b4_s", np.concatenate([O["int_b4_s"], O["outlier_b4_s"]]), float(O["int_b4_s"].max()), True)]
    knob = L.antq_debug_set
    K = 4096
    n_flip_rows = n_rows = 0
    for name, g, gmax, olive in books:
        g = np.ascontiguousarray(g, dtype=np.float32)
        gmax = float(g.max()) if gmax is None else gmax
        plan = antq_lib.plan_for(g)
        hdr = plan.host[:128].view(np.uint32)
        assert plan.is_table and int(hdr[24]) == 3, name
        n_thr, hshifts, tl_off = int(hdr[25]), int(hdr[27]), int(hdr[28])
        T = plan.host[tl_off:tl_off + 16 * n_thr].view(np.float32).reshape(n_thr, 4)[:, 0].copy()
        lim = min(float(plan.host[:128].view(np.float32)[13]) * 0.99999, float(plan.host[:128].view(np.float32)[17]))
        for dtype, tdt in ((1, torch.bfloat16), (2, torch.float16)):
            hshift = (hshifts >> (8 * (dtype - 1))) & 0xff
            inf16 = 0x7f80 if dtype == 1 else 0x7c00
            for ovp in ((False, True) if olive else (False,)):
                flips = _flip_scales(L, plan, gmax, dtype, ovp)
                assert flips, (name, dtype)            # (at the very least: scales too small / too large for any table)
                lo_e, hi_e = (-10.0, 10.0) if dtype == 1 else (-6.5, 5.5)
                alphas = list(np.float32(10.0) ** rng.uniform(lo_e, hi_e, 256).astype(np.float32))
                for a, b in flips[:24]:
                    for v in (a, b):
                        bits = int(_f32_bits(v))
                        alphas += [v, np.uint32(max(bits - 1, 1)).view(np.float32), np.uint32(bits + 1).view(np.float32)]
                alphas = np.float32(alphas)
                alphas = alphas[np.isfinite(alphas) & (alphas > 0)]
                n_flip_rows += len(alphas) - 256
                rows = len(alphas)
                x16 = np.empty((rows, K), np.uint16)
                special = np.uint16([0, 0x8000, inf16, inf16 | 0x8000, inf16 + 1, 0xffff, inf16 - 1, (inf16 - 1) | 0x8000, 1, 0x8001,
                                     2, 0x7fff, inf16 >> 1, 0x3c00 if dtype == 2 else 0x3f80])
                for r, a in enumerate(alphas):
                    with np.errstate(all="ignore"):
                        s = np.float32(a) / np.float32(gmax)
                        u = np.abs(T).astype(np.float32) * s
                        limx = np.float32(min(lim * float(s) * 0.999, 3.0e38))
                    pu = _to16(u[np.isfinite(u) & (u > 0)], dtype).astype(np.int64)
                    pl = int(_to16(np.float32([limx]), dtype)[0]) if limx > 0 else 0
                    cand = [special.astype(np.int64)]
                    for d in range(-3, 4):
                        cand.append(pu + d)
                        cand.append(np.int64([max(pl + d, 0)]))
                    k0 = max((int(pu.min()) >> hshift) - 2, 0) if pu.size else 0
                    k1 = min((pl >> hshift) + 2, (0x7fff >> hshift))
                    if k1 - k0 <= 200:
                        edges = (np.arange(k0, k1 + 1, dtype=np.int64) << hshift)
                        cand += [edges, edges + 1, np.maximum(edges, 1) - 1]
                    mag = np.unique(np.clip(np.concatenate(cand).astype(np.int64), 0, 0x7fff)).astype(np.uint16)
                    both = np.concatenate([mag, mag | 0x8000, special])
                    if both.size > K - 512:
                        both = rng.choice(both, K - 512, replace=False)
                    row = np.concatenate([both, rng.integers(0, 65536, K - both.size).astype(np.uint16)])
                    x16[r] = rng.permutation(row)
                xf = oracle.bf16_to_f32(x16) if dtype == 1 else x16.view(np.float16).astype(np.float32)
                with np.errstate(all="ignore"):
                    ref, _ = oracle.forward(xf, alphas, g, gmax, ovp)
                    ref16 = oracle.f32_to_bf16(ref) if dtype == 1 else ref.astype(np.float16).view(np.uint16)
                rf = oracle.bf16_to_f32(ref16) if dtype == 1 else ref16.view(np.float16).astype(np.float32)
                xt = torch.from_numpy(x16.view(np.int16)).to(dev).view(tdt)
                at = torch.from_numpy(alphas).to(dev)

                def same(t, what):
                    got = t.view(torch.int16).cpu().numpy().view(np.uint16).reshape(ref16.shape)
                    gf = oracle.bf16_to_f32(got) if dtype == 1 else got.view(np.float16).astype(np.float32)
                    bad = ~((got == ref16) | (np.isnan(gf) & np.isnan(rf)))
                    assert not bad.any(), (name, str(tdt), ovp, what, int(bad.sum()), alphas[np.argwhere(bad)[:3, 0]].tolist(),
                                           x16[bad][:3], got[bad][:3], ref16[bad][:3])

                same(antq_lib.fakequant(xt, at, plan, gmax, rows, K, True, ovp=ovp), "one launch")
                knob(0, 8)
                same(antq_lib.fakequant(xt, at, plan, gmax, rows, K, True, ovp=ovp), "one launch, 8-vector tasks")
                knob(0, 0)
                same(antq_lib.fakequant(xt, at, plan, gmax, rows, K, True, ovp=ovp, out=torch.empty_like(xt), unordered=True), "unordered")
                out = torch.empty_like(xt)
                antq_lib.Batch([(xt, out, at, plan, gmax, rows, K, True)], ovp=ovp).run()
                same(out, "batched")
                n_rows += rows
    assert n_flip_rows >= 60 and n_rows > 4000


# ---------------------------------------------------------------------------------------------------------------------------
# the 16-bit-domain encoder (csrc/antq_k_codec.h: k_encode4_hrow)
# ---------------------------------------------------------------------------------------------------------------------------
def _codes_want(oracle, ridx, n_normal, ovp, zero_code):
    want = ridx.astype(np.int64).copy()
    if ovp:
        want[ridx >= n_normal] -= n_normal
    want[ridx == oracle.IDX_VICTIM] = 15
    want[ridx == oracle.IDX_NONE] = zero_code
    return want


def test_16bit_domain_encoder_codes_on_every_pattern(antq_lib, oracle, dev):
    """antq_encode4 on bf16 / f16 rows of >= 128 vectors (k_encode4_hrow: the slot table of K1h holding code bytes; the pair
    rule on the 0x10 flags of four pairs at once): the codes ARE the oracle's scan-order indices (OQ:155-179: an outlier is its
    index in the outlier list, a victim the identifier 15; no entry within 102400: the code of the grid's zero) for EVERY
    one of the 65 536 input patterns, in order and shuffled (other pairs, other lanes), one row per scale -- ordinary ones,
    zero, negative, NaN, Inf, denormal-range, overflowing --, through 8-, 4-, 3- and 2-vector tasks, ANT and OliVe; and equal
    to the fp32-domain row encoder's (knob 9 = 0) and the element encoder's (knob 2 = 0) codes."""
    import torch
    from conftest import golden
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    rng = np.random.default_rng(43)
    allpat = np.arange(65536, dtype=np.uint16)
    books = [("flint_b4_s", G["flint_b4_s"], None, 0), ("int_b4_s", G["int_b4_s"], None, 0), ("flint_b4_u", G["flint_b4_u"], None, 0),
             ("pot_b4_s", G["pot_b4_s"], None, 0), ("int_b3_u", G["int_b3_u"], None, 0),
             ("olive_flint_b4_s", np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]]), float(O["flint_b4_s"].max()), O["flint_b4_s"].size),
             ("olive_int_b4_s", np.concatenate([O["int_b4_s"], O["outlier_b4_s"]]), float(O["int_b4_s"].max()), O["int_b4_s"].size)]
    # (OliVe's UNSIGNED 4-bit codebook has 16 normal values: with the identifier 15 reserved there is no packed form for it --
    #  antq_encode4 refuses it, include/antq.h)
    knob = antq_lib.lib().antq_debug_set
    for name, g, gmax, nn in books:
        g = np.ascontiguousarray(g, dtype=np.float32)
        gmax = float(g.max()) if gmax is None else gmax
        olive = nn > 0
        plan = antq_lib.plan_for(g)
        assert plan.is_table and int(plan.host[:128].view(np.uint32)[24]) == 3, name
        zs = np.flatnonzero((g[:nn] if olive else g) == 0)
        zero_code = int(zs[-1]) if zs.size else 0
        alphas = np.concatenate([np.float32([1.0, 0.06, 0.0, -0.05, np.nan, np.inf, 1e-30, 1e30, 65504.0, 6e-8]),
                                 np.exp(rng.uniform(-20, 20, 6)).astype(np.float32)])
        rows = len(alphas)
        for tdt, npdt in ((torch.bfloat16, None), (torch.float16, np.float16)):
            for pats in (allpat, rng.permutation(allpat)):
                x16 = np.ascontiguousarray(np.broadcast_to(pats, (rows, 65536)))
                xf = oracle.bf16_to_f32(x16) if npdt is None else x16.view(np.float16).astype(np.float32)
                xt = torch.from_numpy(x16.view(np.int16)).to(dev).view(tdt)
                at = torch.from_numpy(alphas).to(dev)
                for ovp in ((True,) if olive else (False,)):
                    with np.errstate(all="ignore"):
                        _, ridx = oracle.forward(xf, alphas, g, gmax, ovp)
                    want = _codes_want(oracle, ridx, nn, ovp, zero_code)

                    def nib(codes, r, k):
                        return torch.stack([(codes & 15), (codes >> 4)], 1).reshape(r, k).cpu().numpy().astype(np.int64)

                    def enc(x, a, r, k):
                        return antq_lib.encode4(x, a, plan, gmax, r, k, True, n_normal=nn, ovp=ovp)

                    base = enc(xt, at, rows, 65536)
                    got = nib(base, rows, 65536)
                    bad = got != want
                    assert not bad.any(), (name, str(tdt), ovp, int(bad.sum()), alphas[np.argwhere(bad)[:3, 0]].tolist(),
                                           x16[bad][:3], got[bad][:3], want[bad][:3])
                    for u in (4, 3, 2):
                        knob(0, u)
                        assert torch.equal(enc(xt, at, rows, 65536), base), (name, str(tdt), u)
                    knob(0, 0)
                    knob(9, 0)
                    assert torch.equal(enc(xt, at, rows, 65536), base), (name, str(tdt), "fp32-domain row encoder")
                    knob(9, 1)
                    # the same elements as rows of 576 vectors (3-vector tasks) and of 128 vectors (2-vector tasks)
                    for rl in (4608, 1024):
                        n_keep = (65536 // rl) * rl
                        xs = xt[:, :n_keep].contiguous().view(-1, rl)
                        a2 = at.repeat_interleave(n_keep // rl)
                        g2 = nib(enc(xs, a2, xs.shape[0], rl), rows, n_keep)
                        assert np.array_equal(g2, want[:, :n_keep]), (name, str(tdt), rl)
    # a per-tensor scale: ONE row however the tensor is shaped
    g = np.ascontiguousarray(G["flint_b4_s"], dtype=np.float32)
    plan = antq_lib.plan_for(g)
    x16 = np.tile(rng.permutation(allpat), 4)
    xt = torch.from_numpy(x16.view(np.int16)).to(dev).view(torch.bfloat16).view(64, 4096)
    with np.errstate(all="ignore"):
        _, ridx = oracle.forward(oracle.bf16_to_f32(x16).reshape(1, -1), np.float32([0.37]), g, 10.0, False)
    codes = antq_lib.encode4(xt, torch.tensor([0.37], device=dev), plan, 10.0, 1, xt.numel(), False)
    got = torch.stack([(codes & 15), (codes >> 4)], 1).reshape(-1).cpu().numpy().astype(np.int64)
    assert np.array_equal(got, _codes_want(oracle, ridx.reshape(-1), 0, False, int(np.flatnonzero(g == 0)[-1])))


# ---------------------------------------------------------------------------------------------------------------------------
# the histogram clip search of 16-bit per-tensor quantisers (csrc/antq_k_hist.h)
# ---------------------------------------------------------------------------------------------------------------------------
def _hist_expected(oracle, x16, dtype, xmax, ratios, grid, gmax):
    """sum_p count[p] * term(p) with the term from the ORACLE's forward on the 65 536 patterns (the reference's fp32 sequence),
    counts from numpy: what the histogram path must produce up to the order of its double additions."""
    pats = np.arange(65536, dtype=np.uint16)
    vals = oracle.bf16_to_f32(pats) if dtype == 1 else pats.view(np.float16).astype(np.float32)
    cnt = np.bincount(x16.reshape(-1).astype(np.int64), minlength=65536).astype(np.float64)
    out = np.empty(ratios.size, np.float64)
    with np.errstate(all="ignore"):
        for c, r in enumerate(ratios):
            a = np.float32(np.float32(xmax) * np.float32(r))
            ref, _ = oracle.forward(vals.reshape(1, -1), np.float32([a]), grid, gmax, False)
            df = np.abs(ref.reshape(-1) - vals).astype(np.float32)
            term = (df * df).astype(np.float32).astype(np.float64)
            out[c] = np.sum(np.where(cnt > 0, cnt * term, 0.0))
    return out


def test_histogram_clip_search_equals_direct_kernels_and_the_oracle(antq_lib, oracle, dev):
    """antq_search_sse / antq_search_sse_multi / antq_calibrate of a bf16 / f16 tensor with ONE scale and no pair rule take the
    histogram path (antq_k_hist.h; knob 14 = 2: for every eligible tensor): the sums equal (a) the oracle's per-pattern terms
    weighted by the tensor's exact pattern counts to 1e-12 -- every element counted once, whatever the tensor's length (whole
    chunks, ragged tails, fewer elements than one chunk), sign mix (ReLU outputs: half the elements on the zero pattern; -0;
    all-negative tensors) and special values (Inf, NaN, denormals) --, (b) the direct kernels' sums (knob 14 = 0) to their
    own fp32 partial-sum noise, with identical clip picks unless the direct scores tie, and (c) each other between the
    single-type and the multi-type entry points, bit for bit.  The whole calibration (antq_calibrate, three codebooks) picks
    the same type and alpha on both paths."""
    import torch
    from conftest import golden
    G = golden("ant_grids.npz")
    rng = np.random.default_rng(91)
    knob = antq_lib.lib().antq_debug_set
    lb, ub = 75, 150
    ratios_np = np.asarray([np.float32(i * 0.01) for i in range(lb, ub)], dtype=np.float32)
    ratios = torch.from_numpy(ratios_np).to(dev)
    names = ["int_b4_s", "pot_b4_s", "flint_b4_s"]
    grids_ = [np.ascontiguousarray(G[k], dtype=np.float32) for k in names]
    plans = [antq_lib.plan_for(g) for g in grids_]
    cases = [("normal", 8 * (1024 * 16 * 4 * 2 + 1024 * 5 + 77), 0), ("relu", 8 * (1024 * 16 * 4 + 3), 1),
             ("tiny", 8 * 300, 0), ("negative", 8 * 1024 * 70, 2), ("specials", 8 * (1024 * 33 + 1), 3), ("big", 1 << 24, 1)]
    try:
        for dtype, tdt in ((1, torch.bfloat16), (2, torch.float16)):
            for cname, n, kind in cases:
                x = (rng.standard_normal(n) * 0.7).astype(np.float32)
                if kind == 1:
                    x = np.maximum(x, 0.0)
                    x[::5] = -0.0
                elif kind == 2:
                    x = -np.abs(x) - 0.01
                elif kind == 3:
                    x[::1001] = np.inf
                    x[7::1003] = 1e-40
                    x[3::997] = -6e4
                x16 = oracle.f32_to_bf16(x) if dtype == 1 else x.astype(np.float16).view(np.uint16)
                xt = torch.from_numpy(x16.view(np.int16)).to(dev).view(tdt)
                vals = oracle.bf16_to_f32(x16) if dtype == 1 else x16.view(np.float16).astype(np.float32)
                fin = vals[np.isfinite(vals)]
                xmax_np = np.float32(np.abs(fin).max())
                xm = torch.tensor([xmax_np], device=dev)
                res = {}
                for mode in (2, 0):
                    knob(14, mode)
                    one = [antq_lib.search_sse(xt, 1, n, xm, False, ratios, p, 10.0) for p in plans]
                    multi = antq_lib.search_sse_multi(xt, 1, n, xm, False, ratios, plans, [10.0] * 3)
                    res[mode] = (torch.stack(one).cpu().numpy().reshape(3, -1), multi.cpu().numpy().reshape(3, -1) if multi is not None else None)
                h_one, h_multi = res[2]
                d_one, _ = res[0]
                assert h_multi is not None and np.array_equal(h_one.view(np.uint64), h_multi.view(np.uint64)), (cname, dtype)
                for t in range(3):
                    if kind != 3:
                        want = _hist_expected(oracle, x16, dtype, xmax_np, ratios_np, grids_[t], 10.0)
                        np.testing.assert_allclose(h_one[t], want, rtol=1e-12, err_msg=str((cname, dtype, names[t])))
                        np.testing.assert_allclose(h_one[t], d_one[t], rtol=2e-6, err_msg=str((cname, dtype, names[t])))
                        ch, cd = int(np.argmin(h_one[t])), int(np.argmin(d_one[t]))
                        assert ch == cd or abs(d_one[t][ch] - d_one[t][cd]) <= 2e-6 * d_one[t][cd], (cname, dtype, names[t], ch, cd)
                    else:                                       # Inf in the tensor: every candidate's sum is NaN on both paths
                        assert np.isnan(h_one[t]).all() and np.isnan(d_one[t]).all(), (cname, dtype)
                if kind != 3:
                    cal = {}
                    for mode in (2, 0):
                        knob(14, mode)
                        alpha, score, typ, xmo = antq_lib.calibrate(xt, 1, n, False, plans, [10.0] * 3, lb, ub, 1, xmax="absmax")
                        cal[mode] = (alpha.cpu().numpy(), score.cpu().numpy(), int(typ.item()), float(xmo.item()))
                    assert cal[2][3] == cal[0][3] == float(xmax_np)
                    np.testing.assert_allclose(cal[2][1], cal[0][1], rtol=2e-6)
                    if cal[2][2] != cal[0][2]:
                        assert abs(cal[0][1][cal[2][2]] - cal[0][1][cal[0][2]]) <= 2e-6 * cal[0][1][cal[0][2]]
                    for t in range(3):
                        if cal[2][0][t, 0] != cal[0][0][t, 0]:
                            ch = int(np.argmin(np.abs(ratios_np * xmax_np - cal[2][0][t, 0])))
                            cd = int(np.argmin(np.abs(ratios_np * xmax_np - cal[0][0][t, 0])))
                            assert abs(d_one[t][ch] - d_one[t][cd]) <= 2e-6 * d_one[t][cd], (cname, dtype, names[t])
        # what the default (knob 14 = 1) does: small tensors keep the direct kernels, large ones switch -- same sums either way
        knob(14, 1)
        n = 1 << 22
        x16 = oracle.f32_to_bf16((rng.standard_normal(n) * 0.3).astype(np.float32))
        xt = torch.from_numpy(x16.view(np.int16)).to(dev).view(torch.bfloat16)
        xm = xt.float().abs().max().reshape(1)
        auto = antq_lib.search_sse_multi(xt, 1, n, xm, False, ratios, plans, [10.0] * 3).cpu().numpy()
        knob(14, 2)
        forced = antq_lib.search_sse_multi(xt, 1, n, xm, False, ratios, plans, [10.0] * 3).cpu().numpy()
        assert np.array_equal(auto.view(np.uint64), forced.view(np.uint64))          # 4 M x 225 evaluations: the histogram pays
        # OliVe pairs, fp32 tensors and per-row quantisers never take it (the sums are those of the direct kernels)
        x32 = torch.randn(1 << 20, device=dev)
        a = antq_lib.search_sse(x32, 1, x32.numel(), x32.abs().max().reshape(1), False, ratios, plans[0], 10.0).cpu().numpy()
        knob(14, 0)
        b = antq_lib.search_sse(x32, 1, x32.numel(), x32.abs().max().reshape(1), False, ratios, plans[0], 10.0).cpu().numpy()
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    finally:
        knob(14, 1)


def test_bench_sharded_workload_line_on_the_gpu(antq_lib, dev):
    """`python bench.py --workload opt6.7b --layers 1` (BASELINE configs[3], shortened to one decoder layer): ONE JSON line,
    strong scaling, the per-rank roofline block, algorithmic bytes = 4 B x the rank's elements, the kernel of the OliVe pairs."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "opt6.7b", "--layers", "1", "--steps", "10",
                          "--warmup", "3", "--no-traffic"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-800:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    elems = 4 * 4096 * 4096 + 2 * 16384 * 4096
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["steps"] == 10 and d["unit"] == "Gelem/s"
    assert d["config"]["elements_per_step_per_rank"] == [elems] and d["config"]["idempotence_check"] is True
    assert "configs[3]" in d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["algorithmic_bytes_per_launch"] == 4 * elems and "k_fq_hbatch<bf16,true>" in r["kernel"]
    assert r["per_rank"]["ranks"] == 1 and r["per_rank"]["frac"]["min"] == r["per_rank"]["frac"]["max"] == r["frac"]
    assert abs(d["value"] - elems / (d["ms_per_step"] * 1e-3) / 1e9) < 0.01 * d["value"] and 0.3 < r["frac"] < 1.0


def test_multihead_attention_options_vs_reference_fixture(antq_lib, dev, capsys):
    """N4 / SURVEY row 7: MultiheadAttentionQuantizer with add_bias_kv, add_zero_attn and both (batch_first), each rewritten,
    calibrated and run by the REFERENCE (tests/golden/ant_mha_options.npz, make_golden.py --tree ant_mha); its checkpoint is
    loaded here (strict: same parameter names incl. bias_k / bias_v), the four quantisers reproduce the recorded tensors bit for
    bit, output and attention weights (one more key column per option) agree within softmax / GEMM rounding."""
    import torch
    import torch.nn as nn
    from conftest import golden
    from test_gpu_parity import _check_recorded_quantizers, _ref_checkpoint
    from ant_quantization_amd.ant import quant_model as qmod, quant_utils as qutil
    from ant_quantization_amd.ant.multihead_attention import MultiheadAttentionQuantizer
    fx = golden("ant_mha_options.npz")
    qutil.set_quantizer(_args(mode="ant-int-pot-flint", wbit=4, abit=4))

    def weight_of(model, qname):
        mha = model.get_submodule(qname.rsplit(".", 1)[0])
        return mha.in_proj_weight if "in_quant" in qname else mha.out_proj_weight

    for pre, kw, cols in (("mk__", dict(add_bias_kv=True), 11), ("mz__", dict(add_zero_attn=True), 11),
                          ("mkz__", dict(add_bias_kv=True, add_zero_attn=True, batch_first=True), 12)):
        torch.manual_seed(7)
        model = qmod.quantize_model(nn.Sequential(nn.MultiheadAttention(64, 4, **kw))).to(dev).eval()
        assert type(model[0]) is MultiheadAttentionQuantizer
        qutil.enable_quantization(model)
        check = _ref_checkpoint(fx, [pre], dev, True)
        qmod.load_ant_state_dict(model, check)
        model.load_state_dict(check, strict=True)
        capsys.readouterr()
        _check_recorded_quantizers(fx, pre, model, dev, weight_of)
        x = torch.from_numpy(fx[pre + "x"]).to(dev)
        with torch.no_grad():
            y, w = model[0](x, x, x)
        assert "-bit" not in capsys.readouterr().out and w.shape[-1] == cols
        np.testing.assert_allclose(y.cpu().numpy(), fx[pre + "y"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(w.cpu().numpy(), fx[pre + "attn_weights"], rtol=2e-4, atol=2e-6)


def test_fused_float64_forward_equals_the_composed_op_sequence(antq_lib, oracle, dev):
    """antq_fakequant_f64 (one kernel) against core.fake_quant_f64's composed path (the reference's seven double ops around the
    float-narrowing operator -- itself pinned to the oracle in test_float64_forward_is_the_reference_sequence_...), which a call
    that wants gradients still takes: same bits, per channel and per tensor, ANT and OliVe pairs, odd element counts (the
    wrap of the last element), Inf / NaN / huge values, zero / negative / NaN alphas."""
    import torch
    from conftest import golden
    from ant_quantization_amd import core
    G, O = golden("ant_grids.npz"), golden("olive_grids.npz")
    gol = np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]])
    rng = np.random.default_rng(65)
    for shape in ((16, 256), (7, 33), (5, 3, 7, 7), (1, 4097), (33, 1)):
        x = rng.standard_normal(shape) * 0.05
        f = x.reshape(-1)
        f[::19] *= 25
        if f.size > 40:
            f[3], f[8], f[11], f[12] = np.inf, np.nan, 1e300, -1e-310
        xt = torch.from_numpy(x).to(dev)
        for g, gmax, ovp in ((G["flint_b4_s"], 10.0, False), (G["int_b8_s"], 10.0, False), (gol, 32.0, True)):
            plan = antq_lib.plan_for(np.ascontiguousarray(g, dtype=np.float32))
            for per_channel in (True, False):
                na = shape[0] if per_channel else 1
                alpha = torch.from_numpy(np.abs(rng.standard_normal(na)) * 0.1 + 0.01).to(dev)
                if na > 4:
                    alpha[1], alpha[2], alpha[3] = 0.0, -0.05, float("nan")
                a = alpha.reshape(-1, 1) if per_channel else alpha.reshape(())
                with torch.no_grad():
                    fused = core.fake_quant(xt, a, plan, gmax, per_channel, ovp=ovp)
                xg = xt.clone().requires_grad_(True)
                composed = core.fake_quant(xg, a, plan, gmax, per_channel, ovp=ovp)
                assert composed.requires_grad and fused.dtype == torch.float64 and fused.shape == xt.shape
                fb, cb = fused.cpu().numpy(), composed.detach().cpu().numpy()
                same = (fb.view(np.uint64) == cb.view(np.uint64)) | (np.isnan(fb) & np.isnan(cb))
                assert same.all(), (shape, ovp, per_channel, int((~same).sum()), fb[~same][:3], cb[~same][:3])


def test_histogram_clip_search_with_outlier_victim_pairs(antq_lib, oracle, dev):
    """The histogram clip search under OliVe's pair rule (OQ:311-320): sum_p count[p] term(p) plus the victims' corrections
    from the list of outlier-capable pairs.  Against (a) the ORACLE's forward with the pair rule on the whole tensor, per
    candidate (terms fl32(|out - x|^2), added in double): 1e-9 -- every victim found, none invented, whatever the pair is
    made of (outlier + normal, two outliers, a value beyond the scan's horizon -- q = 0, not an outlier --, a zero partner); (b) the direct kernels (knob 14 = 0):
    their own partial-sum noise, same picks unless they tie; (c) itself: five runs, identical bits (the list's layout does not
    depend on which wavefront finished first).  A tensor whose pair list overflows falls back to the direct kernels on the
    device: bit-identical sums, no error, no synchronisation."""
    import torch
    from conftest import golden
    O = golden("olive_grids.npz")
    rng = np.random.default_rng(92)
    knob = antq_lib.lib().antq_debug_set
    books = [(np.concatenate([O["flint_b4_s"], O["outlier_b4_s"]]).astype(np.float32), float(O["flint_b4_s"].max())),
             (np.concatenate([O["int_b4_s"], O["outlier_b4_s"]]).astype(np.float32), float(O["int_b4_s"].max()))]
    plans = [antq_lib.plan_for(g) for g, _ in books]
    gmaxs = [gm for _, gm in books]
    ratios_np = np.asarray([np.float32(i * 0.01) for i in range(80, 250, 10)], dtype=np.float32)
    ratios = torch.from_numpy(ratios_np).to(dev)
    try:
        for dtype, tdt in ((1, torch.bfloat16), (2, torch.float16)):
            for cname, n, frac in (("sparse outliers", 8 * (1024 * 16 * 4 + 1024 * 3 + 5), 0.004), ("many outliers", 1 << 18, 0.05),
                                   ("specials", 8 * 1024 * 40, 0.003)):
                x = (rng.standard_normal(n) * 0.05).astype(np.float32)
                idx = rng.choice(n, int(n * frac), replace=False)
                x[idx] *= rng.uniform(6, 60, idx.size).astype(np.float32)
                x[idx[:20] ^ 1] = x[idx[:20]] * 1.5                              # pairs of two outliers
                xmax_np = np.float32(3.0 * x.std())                              # (the clip statistic is an input here)
                if cname == "specials":
                    x[idx[20:24]] = 3e4                                          # beyond the scan's horizon: q = 0, NOT an outlier
                    x[idx[30] ^ 1] = 0.0
                x16 = oracle.f32_to_bf16(x) if dtype == 1 else x.astype(np.float16).view(np.uint16)
                vals = oracle.bf16_to_f32(x16) if dtype == 1 else x16.view(np.float16).astype(np.float32)
                xt = torch.from_numpy(x16.view(np.int16)).to(dev).view(tdt)
                xm = torch.tensor([xmax_np], device=dev)
                res = {}
                for mode in (2, 0):
                    knob(14, mode)
                    one = [antq_lib.search_sse(xt, 1, n, xm, False, ratios, p, gm, ovp=True) for p, gm in zip(plans, gmaxs)]
                    multi = antq_lib.search_sse_multi(xt, 1, n, xm, False, ratios, plans, gmaxs, ovp=True)
                    res[mode] = (torch.stack(one).cpu().numpy().reshape(2, -1), multi.cpu().numpy().reshape(2, -1))
                h_one, h_multi = res[2]
                d_one, d_multi = res[0]
                # (single- and multi-type calls list the outlier-capable pairs from different bounds -- the lowest threshold of
                #  the codebooks searched together -- so the victims' corrections are added in another order, or one of the two
                #  overflows its list and is answered by the direct kernels: equal to rounding, not bit for bit)
                np.testing.assert_allclose(h_one, h_multi, rtol=1e-11 if frac < 0.01 else 2e-6, err_msg=str((cname, dtype)))
                for t, (g, gm) in enumerate(books):
                    want = np.empty(ratios_np.size)
                    with np.errstate(all="ignore"):
                        for c, r in enumerate(ratios_np):
                            a = np.float32(np.float32(xmax_np) * np.float32(r))
                            ref, ridx = oracle.forward(vals.reshape(1, -1), np.float32([a]), g, gm, True)
                            df = np.abs(ref.reshape(-1) - vals).astype(np.float32)
                            want[c] = float(np.sum((df * df).astype(np.float32).astype(np.float64)))
                            if c == 0:
                                assert (ridx == oracle.IDX_VICTIM).sum() > 10, (cname, dtype)
                    np.testing.assert_allclose(h_one[t], want, rtol=1e-9 if frac < 0.01 else 2e-6, equal_nan=True, err_msg=str((cname, dtype, t)))
                    np.testing.assert_allclose(h_multi[t], want, rtol=1e-9 if frac < 0.01 else 2e-6, equal_nan=True, err_msg=str((cname, dtype, t)))
                    np.testing.assert_allclose(d_one[t], want, rtol=2e-6, equal_nan=True, err_msg=str((cname, dtype, t, "direct")))
                    if np.isfinite(want).all():
                        ch, cd = int(np.argmin(h_one[t])), int(np.argmin(d_one[t]))
                        assert ch == cd or abs(d_one[t][ch] - d_one[t][cd]) <= 2e-6 * d_one[t][cd]
                knob(14, 2)
                for _ in range(4):
                    again = antq_lib.search_sse_multi(xt, 1, n, xm, False, ratios, plans, gmaxs, ovp=True).cpu().numpy().reshape(2, -1)
                    assert np.array_equal(again.view(np.uint64), h_multi.view(np.uint64)), (cname, dtype, "not reproducible")
        # overflow of the pair list: a clip statistic of ONE sigma makes a third of the elements outlier-capable -> the direct
        # kernels answer, same bits
        n = 1 << 20
        x = (rng.standard_normal(n) * 0.05).astype(np.float32)
        x16 = oracle.f32_to_bf16(x)
        xt = torch.from_numpy(x16.view(np.int16)).to(dev).view(torch.bfloat16)
        xm = torch.tensor([np.float32(0.05)], device=dev)
        knob(14, 2)
        over = antq_lib.search_sse_multi(xt, 1, n, xm, False, ratios, plans, gmaxs, ovp=True).cpu().numpy()
        over1 = antq_lib.search_sse(xt, 1, n, xm, False, ratios, plans[0], gmaxs[0], ovp=True).cpu().numpy()
        knob(14, 0)
        direct = antq_lib.search_sse_multi(xt, 1, n, xm, False, ratios, plans, gmaxs, ovp=True).cpu().numpy()
        direct1 = antq_lib.search_sse(xt, 1, n, xm, False, ratios, plans[0], gmaxs[0], ovp=True).cpu().numpy()
        assert np.isfinite(over).all() and np.array_equal(over.view(np.uint64), direct.view(np.uint64))
        assert np.array_equal(over1.view(np.uint64), direct1.view(np.uint64))
    finally:
        knob(14, 1)


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_graph_of_the_default_schedule_requantises_on_every_replay(antq_lib, dev, tree):
    """Default schedule: the forward's ONE batched weight launch is part of the captured graph, so a replay after ANY weight
    change -- an in-place update or a raw `.data` edit -- computes what the reference's cache-nothing forward would
    (AQ:613-617), with nothing to call in between."""
    import torch
    import torch.nn as nn
    qmod, qutil = _trees(tree)
    qutil.set_quantizer(_args(mode="flint", wbit=4, abit=4))
    torch.manual_seed(4)
    net = nn.Sequential(nn.Linear(128, 256), nn.GELU(), nn.Linear(256, 256), nn.GELU(), nn.Linear(256, 32))
    model = qmod.quantize_model(net).to(dev).eval()
    qutil.enable_quantization(model)
    static_x = torch.randn(32, 128, device=dev)
    with torch.no_grad():
        model(static_x)                                  # calibration
        model(static_x)                                  # the bank attaches
    bank = model._antq_auto_bank.bank
    assert bank is not None and not bank.resident
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        model(static_x)
    torch.cuda.current_stream().wait_stream(side)
    n = bank.launches
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        static_y = model(static_x)
    assert bank.launches == n + 1                        # the refresh was captured
    for step in range(3):
        for p in model.parameters():
            if p.dim() == 2:
                p.data.mul_(1.0 + 0.05 * (step + 1))     # no version counter moves, nothing is told
        xn = torch.randn(32, 128, device=dev, generator=torch.Generator(device=dev).manual_seed(step))
        static_x.copy_(xn)
        graph.replay()
        torch.cuda.synchronize()
        ref_model = copy.deepcopy(model)
        qutil.set_weight_bank(ref_model, False)          # the reference's schedule: every layer re-quantises its weight
        with torch.no_grad():
            ref = ref_model(xn)
        assert torch.equal(static_y, ref), (tree, step)
    assert bank.launches == n + 1                        # (replays launch from the graph, not from Python)


@pytest.mark.parametrize("tree", ["ant", "olive"])
def test_graph_replay_after_a_weight_change_with_the_bank_refreshed_outside(antq_lib, dev, tree):
    """INTEGRATION.md, resident mode (set_weights_at_rest): a graph captured on clean resident weights holds no weight launch; after a weight change ONE
    `bank.refresh()` outside the graph (the resident buffers keep their addresses) makes the replay equal to the eager
    forward of the changed model -- which the reference's cache-nothing schedule (AQ:613-617) would compute."""
    import torch
    import torch.nn as nn
    qmod, qutil = _trees(tree)
    qutil.set_quantizer(_args(mode="flint", wbit=4, abit=4))
    torch.manual_seed(3)
    net = nn.Sequential(nn.Linear(128, 256), nn.GELU(), nn.Linear(256, 256), nn.GELU(), nn.Linear(256, 32))
    model = qmod.quantize_model(net).to(dev).eval()
    qutil.enable_quantization(model)
    qutil.set_weights_at_rest(model, True)               # opt-in: resident copies, no weight launch while nothing changes
    static_x = torch.randn(32, 128, device=dev)
    with torch.no_grad():
        model(static_x)                                  # calibration
        model(static_x)                                  # the bank attaches and fills
    bank = model._antq_auto_bank.bank
    assert bank is not None and bank.launches == 1 and bank.resident
    addrs = [e["out"].data_ptr() for e in bank.entries.values()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        model(static_x)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        static_y = model(static_x)
    assert bank.launches == 1                            # nothing launched for the weights inside the capture
    for step in range(3):
        with torch.no_grad():
            for p in model.parameters():
                if p.dim() == 2:
                    p.mul_(1.0 + 0.05 * (step + 1))      # a weight update between two replays
        bank.refresh()                                   # one launch, outside the graph
        assert [e["out"].data_ptr() for e in bank.entries.values()] == addrs
        xn = torch.randn(32, 128, device=dev, generator=torch.Generator(device=dev).manual_seed(step))
        static_x.copy_(xn)
        graph.replay()
        torch.cuda.synchronize()
        ref_model = copy.deepcopy(model)                 # (a copy starts without a bank; the graph's bank stays attached)
        qutil.set_weight_bank(ref_model, False)          # the reference's schedule: every layer re-quantises its weight
        with torch.no_grad():
            ref = ref_model(xn)
        assert torch.equal(static_y, ref), (tree, step)
        assert model._antq_auto_bank.bank is bank


def test_calibrate_one_scale_absmax_from_the_counting_pass_and_one_pick_launch(antq_lib, dev):
    """antq_calibrate of a tensor with ONE scale (every activation quantiser, AQ:308 / :328-415): (a) on the histogram path
    the abs-max statistic comes out of the counting pass (knob 15 = 1, the default) -- same x_max, alphas, scores and type,
    bit for bit, as with the separate abs-max pass (knob 15 = 0), whatever the length (whole chunks, ragged tails, less than
    one chunk), sign mix and special values (Inf, NaN: x_max NaN like torch.max); (b) the per-type pick, the type's score and
    the type pick share one launch: equal to antq_search_pick per type on the same sums and to argsort(score)[0] (first
    minimum, NaN last) -- for fp32 tensors (direct kernels) as well."""
    import torch
    from ant_quantization_amd import grids
    knob = antq_lib.lib().antq_debug_set
    rng = np.random.default_rng(17)
    plans = [antq_lib.plan_for(grids.ant_grid(t, 4, True)) for t in ("int", "flint", "pot")]
    gm = [10.0] * 3
    lb, ub = 75, 150
    ratios = torch.from_numpy(np.asarray([np.float32(i * 0.01) for i in range(lb, ub)], dtype=np.float32)).to(dev)

    def bits(t):
        return t.contiguous().view(torch.int32).cpu().numpy()

    cases = [("normal", 8 * (1024 * 16 * 4 * 2 + 1024 * 5 + 77), 0), ("relu", 8 * (1024 * 16 * 4 + 3), 1), ("tiny", 8 * 300, 0),
             ("negative", 8 * 1024 * 70, 2), ("inf", 8 * (1024 * 33 + 1), 3), ("nan", 8 * (1024 * 9 + 5), 4), ("big", 1 << 23, 0),
             ("max in the last lane", 8 * (1024 * 16 + 1), 5)]
    try:
        for tdt in (torch.bfloat16, torch.float16, torch.float32):
            for cname, n, kind in cases:
                x = (rng.standard_normal(n) * 0.7).astype(np.float32)
                if kind == 1:
                    x = np.maximum(x, 0.0)
                elif kind == 2:
                    x = -np.abs(x) - 0.01
                elif kind == 3:
                    x[rng.integers(0, n, 3)] = np.inf
                elif kind == 4:
                    x[rng.integers(0, n, 2)] = np.nan
                    x[7] = np.inf
                elif kind == 5:
                    x[-1] = -37.5                        # the maximum sits in the ragged tail, negative
                xt = torch.from_numpy(x).to(dev).to(tdt)
                res = {}
                for k15 in (0, 1):
                    knob(14, 2)                          # the histogram path for every eligible tensor (16-bit ones)
                    knob(15, k15)
                    res[k15] = antq_lib.calibrate(xt, 1, n, False, plans, gm, lb, ub, 1)
                a0, s0, t0, m0 = res[0]
                a1, s1, t1, m1 = res[1]
                tag = (str(tdt), cname)
                assert np.array_equal(bits(m0), bits(m1)), tag
                assert np.array_equal(bits(a0), bits(a1)) and np.array_equal(bits(s0), bits(s1)) and int(t0) == int(t1), tag
                ref = xt.float().abs().max()
                if kind == 4:
                    assert bool(torch.isnan(m1).all()), tag
                else:
                    assert float(m1) == float(ref), tag
                # (b) the pick launch against the public per-type pick on the same sums
                sse = antq_lib.search_sse_multi(xt, 1, n, m1, False, ratios, plans, gm)          # [types, candidates]
                sc = []
                for t in range(3):
                    best, al = antq_lib.search_pick(sse[t].reshape(-1, 1).contiguous(), m1, ratios, n)
                    assert np.array_equal(bits(best), bits(s1[t:t + 1])), (tag, t)
                    assert np.array_equal(bits(al), bits(a1[t])), (tag, t)
                    sc.append(float(best))
                order = [t for t in range(3) if sc[t] == sc[t]]
                want = min(order, key=lambda t: sc[t]) if order else 0
                assert int(t1) == want, (tag, sc, int(t1))
    finally:
        knob(14, 1)
        knob(15, 1)
