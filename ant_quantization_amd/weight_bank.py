"""One launch for every calibrated weight quantiser of a model.

The reference re-quantises each layer's weight on every forward, one small kernel sequence per layer
(AQ:608-611, :641-644; OQ:413-416, :443-446).  For ResNet-50 that is 54 launches of a few tens of KB to
a few MB each -- launch-bound, 7 % of the HBM roofline here -- while the same bytes through ONE
multi-tensor launch (`antq_fakequant_batch`) reach 75 %, and 80 % for LLM-sized tensors.

`WeightBank(model)` collects the steady-state weight quantisers of the wrapper layers, quantises all
their weights in one launch into bank-owned buffers, and hands each layer its buffer on forward.  The
result is the same tensor the per-layer kernel would have produced (same kernels underneath, parity
tested bit-exact).  Two schedules:

  * DEFAULT (what `enable_quantization(model)` arms, AutoBank): the reference's schedule in one launch.  Nothing is
    cached across forwards: EVERY forward that does not need gradients through the quantisers re-quantises every weight
    -- one batched launch (0.16 ms for BERT-base), issued by the first weight quantiser the forward reaches, capturable
    into a hipGraph (a replayed graph re-quantises too).  "A new forward" = the model's forward returned since the last
    refresh (forward hook), or a layer asks for its weight a second time since then (layers called directly, a module
    used twice in one forward), or a stamp (address / version of weight and alpha, codebook) moved.  So a raw
    `w.data.mul_()` between two no-grad forwards is seen, exactly as in the reference (AQ:613-617, :642-646).
  * RESIDENT (`quant_utils.set_weights_at_rest(model, True)`, or `WeightBank(model)` built by hand): inference on frozen
    weights.  The quantised copies stay; a forward on unchanged weights launches NOTHING for them.  Weights and alphas
    are stamped by (`data_ptr`, `_version`); a stale stamp, a training forward, a `train()` / `.eval()` call on a wrapped
    layer (optimisers that write through `.data`, like the reference's BertAdam, BERT/optimization.py:161) all trigger
    one refresh at the next no-grad forward.  What this mode cannot see is a raw `.data` edit with none of those around
    it -- which is what "at rest" promises does not happen; `bank.invalidate()` says it did.

In both, a forward that needs gradients through the quantiser bypasses the bank (autograd path, AQ:544-549).

Costs: one bank-owned quantised copy of every weight, and the tensor a layer receives is that buffer (rewritten at the
next refresh) rather than a fresh one.  `quant_utils.set_weight_bank(model, False)` (or ANTQ_WEIGHT_BANK=0 in the
environment) restores the reference's per-layer launches exactly.
"""
import os
import warnings
import weakref

import torch

from . import _lib
from ._mirror import _lib_tensor_key as _tensor_key

__all__ = ["WeightBank", "AutoBank", "FlushHook"]


def _weight_layers(model):
    for name, mod in model.named_modules():
        q = getattr(mod, "quant_weight", None)
        w = getattr(mod, "weight", None)
        if q is not None and isinstance(w, torch.Tensor) and hasattr(q, "tensor_forward"):
            yield name, mod, q, w


class _LazyPick:
    """A type pick on its way to pinned memory: `get()` waits for the copy of ITS calibrate_batch call, nobody else's."""
    __slots__ = ("ev", "host", "k")

    def __init__(self, ev, host, k):
        self.ev, self.host, self.k = ev, host, k

    def __deepcopy__(self, memo):
        return self

    def get(self):
        self.ev.synchronize()
        return int(self.host[self.k])


class WeightBank:
    def __init__(self, model, resident=True):
        self.model = model
        self.resident = bool(resident)   # False: the reference's schedule (module docstring), what AutoBank builds by default
        self._served = set()       # id(quantiser) handed its buffer since the last refresh (non-resident schedule)
        self.entries = {}          # id(quantiser) -> dict
        self._batches = []
        self._ptr_key = None
        self.dirty = True          # set by whatever may have changed weights behind the stamps' back (mark_dirty)
        self.launches = 0          # refreshes run so far (for tests / logging)
        self.skipped = []          # (layer name, reason) of layers the bank leaves to the per-layer path
        self.auto = None           # weakref to the AutoBank that built this bank (told when the bank gives up)
        try:
            for name, mod, q, w in _weight_layers(model):
                reason = self._unsuitable(q, w)
                if reason:
                    self.skipped.append((name, reason))
                    continue
                rows, row_len = (w.shape[0], w.numel() // w.shape[0]) if q.is_perchannel else (1, w.numel())
                self.entries[id(q)] = dict(name=name, q=q, mod=mod, rows=rows, row_len=row_len, w=w, a=None, plan=None, gmax=None,
                                           out=torch.empty_like(w, memory_format=torch.contiguous_format), stamp=None)
                q._bank = self
        except BaseException:      # (out of memory half way through: leave nothing attached)
            self.detach()
            raise
        if not self.entries:
            raise _lib.AntqError("WeightBank: no calibrated weight quantiser found (run one forward to calibrate first)")

    def __deepcopy__(self, memo):       # (a copy of the model starts without a bank; FlushHook re-arms an AutoBank for it)
        return None

    def __reduce__(self):
        return (type(None), ())

    @staticmethod
    def _unsuitable(q, w):
        if q.mode in ("base", "outlier"):
            return "mode %s" % q.mode
        if not (q.is_enable and q.is_enable_weight):
            return "quantisation disabled"
        if not q._steady:
            return "not calibrated yet"
        if not w.is_cuda or not w.is_contiguous():
            return "weight not resident / not contiguous"
        if w.dtype not in _lib._DTYPES or w.dtype == torch.float64:
            return "dtype %s" % w.dtype
        return None

    # ------------------------------------------------------------------ bookkeeping
    @staticmethod
    def _stamp(q, w):
        a = q.alpha
        return (w.data_ptr(), w._version, a.data_ptr(), a._version, id(q._plan), q._gmax)

    def invalidate(self):
        """The next lookup re-quantises every weight (one launch).  For edits the stamps cannot see: `.data` writes."""
        self.dirty = True

    mark_dirty = invalidate

    def nbytes(self):
        return sum(e["out"].numel() * e["out"].element_size() for e in self.entries.values())

    def detach(self):
        for e in self.entries.values():
            e["q"]._bank = None
        self.entries.clear()
        self._batches = []

    def _build(self):
        """(Re)build the descriptor tables: one batch per (dtype, victim-pairs on/off)."""
        groups = {}
        keep = []
        for e in self.entries.values():
            q, w = e["q"], e["mod"].weight
            if e["out"].dtype != w.dtype or e["out"].device != w.device or e["out"].shape != w.shape:
                e["out"] = torch.empty_like(w, memory_format=torch.contiguous_format)   # .half() / .to(device) since
            plan = q._ensure_plan()
            alpha = q.alpha.detach().reshape(-1)
            if alpha.dtype != torch.float32 or not alpha.is_contiguous():
                alpha = alpha.to(torch.float32).contiguous()
                e["alpha_copy"] = True       # stamps then also cover the copy: rebuilt whenever alpha changes
            else:
                e["alpha_copy"] = False
            keep.append(alpha)
            ovp = bool(getattr(q, "_no_outlier", True) is False)
            groups.setdefault((w.device, w.dtype, ovp), []).append(
                (w.detach(), e["out"], alpha, plan, q._gmax, e["rows"], e["row_len"], bool(q.is_perchannel)))
        self._batches = [_lib.Batch(jobs, ovp=ovp) for (_, _, ovp), jobs in groups.items()]
        self._keep = keep
        self._ptr_key = self._pointers()
        # what lookup() compares per layer (default schedule) and the compiled refresh: the tensors whose addresses the
        # tables hold, and those addresses
        ts, vs = [], []
        for e in self.entries.values():
            e["w"], e["a"], e["plan"], e["gmax"] = e["mod"].weight, e["q"]._parameters.get("alpha"), e["q"]._plan, e["q"]._gmax
            ts += [e["w"], e["q"].alpha]
            vs += [-1, -1]
            for g in (e["q"].quant_grid, getattr(e["q"], "outliers", None)):      # (the codebook buffers: an in-place edit
                if isinstance(g, torch.Tensor):                                   #  must rebuild the plan)
                    ts.append(g)
                    vs.append(-1 if torch.is_inference(g) else g._version)
        self._fast = None
        if _lib.ext() is not None and hasattr(_lib.ext(), "bank_refresh") and not any(b.singles for b in self._batches) \
                and not any(e["alpha_copy"] for e in self.entries.values()):
            self._fast = (ts, [t.data_ptr() for t in ts], vs,
                          [(b.host.ctypes.data, b.dev) for b in self._batches if b.host is not None])

    def _pointers(self):
        return tuple((e["mod"].weight.data_ptr(), e["mod"].weight.dtype, e["q"].alpha.data_ptr(), id(e["q"]._ensure_plan()),
                      e["q"]._gmax) for e in self.entries.values())

    # ------------------------------------------------------------------ the one launch
    def _refresh_fast(self):
        """Default schedule, every forward: one compiled call checks that no weight / alpha moved and launches the batches
        (~20 us of host time for BERT-base's 73 layers; refresh() below walks them in Python: ~0.4 ms).  A codebook change
        is caught per layer in lookup(); version counters do not matter here -- every forward re-quantises anyway."""
        f = self._fast
        if f is None or not _lib.ext().bank_refresh(f[0], f[1], f[2], f[3]):
            return False
        self.launches += 1
        self.dirty = False
        self._served.clear()
        return True

    @torch.no_grad()
    def refresh(self):
        need_build = self._ptr_key != self._pointers() or any(e.get("alpha_copy") for e in self.entries.values())
        if need_build or not self._batches:
            self._build()
        for b in self._batches:
            b.run()
        self.launches += 1
        self.dirty = False
        self._served.clear()
        for e in self.entries.values():
            e["stamp"] = self._stamp(e["q"], e["mod"].weight)

    def lookup(self, q, tensor, training=None):
        """Called from TensorQuantizer.tensor_forward in steady state.  Returns the resident fake-quantised
        weight, or None when this forward has to take the per-layer path.  `training`: the caller's forward wants gradients
        for the weight or alpha (looked at before tensor_forward entered no_grad, OQ:332); None = decide here."""
        e = self.entries.get(id(q))
        if e is None:
            return None
        if tensor is not e["w"]:
            if tensor is not e["mod"].weight:
                return None                    # (the quantiser applied to some other tensor: its own launch)
            e["w"] = tensor                    # the layer's Parameter object was replaced: new tables
            self._fast = self._ptr_key = None
            self.dirty = True
        if training is None:
            training = torch.is_grad_enabled() and (tensor.requires_grad or q.alpha.requires_grad)
        if training:
            # a training forward: an optimiser step follows, and it may write through `.data` without moving any version
            # counter (BERT/optimization.py:161) -- nothing resident is trusted after this
            self.dirty = True
            return None
        if not q._steady or not (q.is_enable and q.is_enable_weight):
            return None
        try:
            if self.resident:
                # only what the stamps / dirty flag say has changed
                if self.dirty or e["stamp"] != self._stamp(q, tensor):
                    self.refresh()
            else:
                # the reference's schedule: a new forward has begun (module docstring) -> every weight again, one launch
                if q._plan is not e["plan"] or q._gmax != e["gmax"] or q._parameters.get("alpha") is not e["a"]:
                    self._fast = None          # (a new codebook / alpha Parameter for this layer: tables rebuilt by refresh())
                    self.dirty = True
                if self.dirty or id(q) in self._served:
                    if not self._refresh_fast():
                        self.refresh()
                self._served.add(id(q))
        except torch.cuda.OutOfMemoryError:
            self._give_up("out of memory while refreshing the bank's weights")
            return None
        return e["out"]

    def _give_up(self, why):
        """The bank cannot serve this model (memory): detach for good, the per-layer schedule takes over."""
        auto = self.auto() if self.auto is not None else None
        self.detach()
        if auto is not None:
            auto.bank = None
            auto.enabled = False
            auto.reason = why
        warnings.warn("ant_quantization_amd: weight bank switched off (%s); weights are quantised per layer" % why)


class AutoBank:
    """Armed by enable_quantization(model): attaches a WeightBank the first time a forward starts with every weight
    quantiser calibrated.  Holds the model weakly; never copied or pickled with the quantisers that point at it."""

    def __init__(self, model):
        self._model = weakref.ref(model)
        self.bank = None
        self.enabled = os.environ.get("ANTQ_WEIGHT_BANK", "1") != "0"
        self.resident = False      # quant_utils.set_weights_at_rest(model, True): the bank keeps its copies across forwards
        self.first = None          # the quantiser whose forward comes first in registration order: the only one that pokes
        self.failed = 0
        self.reason = None         # why the bank was switched off by itself, if it was (memory)
        # the resident copies may take at most this fraction of the device memory that is free when the bank is built
        self.mem_fraction = float(os.environ.get("ANTQ_BANK_MEM_FRACTION", "0.25"))
        self.batch_calibration = {"0": 0, "2": 2}.get(os.environ.get("ANTQ_BATCH_CALIB", "1"), 1)
        self.defer_types = os.environ.get("ANTQ_DEFER_TYPES", "1") != "0"      # _mirror.CalibrationMixin._calibrate_deferred
        self.queue = []            # calibration log: lines in order, quantisers whose line waits for its type pick
        self.deferred = 0          # picks made on the device so far (tests / logging)
        for mod in model.modules():
            for q in (getattr(mod, "quant_weight", None), getattr(mod, "quant_input", None)):
                if q is not None and hasattr(q, "_calib_ctx"):
                    q._calib_ctx = self
        self.precal_done = False
        self.precalibrated = 0     # weight quantisers calibrated by precalibrate() (tests / logging)
        for _, _, q, _ in _weight_layers(model):
            if self.first is None:
                self.first = weakref.ref(q)
            q._auto_bank = self
            self.resident = self.resident or bool(getattr(q, "weights_at_rest", False))    # (set_weights_at_rest came first)

    def __deepcopy__(self, memo):
        return None

    # ------------------------------------------------------------------ the calibration log, in the reference's order
    def emit(self, line):
        if self.queue:
            self.queue.append((None, line))
        elif line:
            print(line)

    def defer(self, q):
        self.queue.append((q, None))
        self.deferred += 1

    def flush(self):
        """End of the model's forward (forward hook), or a pending quantiser's next forward: learn every pending pick -- they
        arrived long ago -- and print the lines that queued up behind them."""
        queue, self.queue = self.queue, []
        for q, line in queue:
            if q is not None:
                line = q._resolve_pending() if q._pending is not None else None
            if line:
                print(line)

    # ------------------------------------------------------------------ the weights' calibration, all at once
    def precalibrate(self):
        """From the first not-yet-calibrated weight quantiser whose forward runs: search EVERY weight quantiser of the model
        now, in one call (antq_calibrate_batch), and read all type picks back in one copy.  A weight's calibration depends on
        nothing but the weight (AQ:468-533 with data = self.weight), so each quantiser gets the state its own first forward
        would have given it -- same kernels, same bits; it installs that state (and prints the reference's line) when its
        forward comes, while the stream is busy with the activations' searches.  What this saves is the read-back per type
        selection (np.argsort(...cpu()), AQ:413): 73 stream drains for BERT-base's weights become one.  Models whose weight
        quantisers select no type (a fixed mode: their per-layer calibration is sync-free already and overlaps the stream),
        quantisers the batch cannot take (outlier / base modes, float1-4 types, an empty candidate range, a float64 or host
        weight) and runs under torch.distributed (the reference's per-quantiser collectives keep their order) take the
        per-layer path as before.  ANTQ_BATCH_CALIB=0 switches this off, =2 batches fixed modes too."""
        if self.precal_done or not self.batch_calibration:
            return
        self.precal_done = True
        model = self._model()
        if model is None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            return
        groups = {}
        with torch.no_grad():
            for name, mod, q, w in _weight_layers(model):
                spec = q._calib_spec(w) if hasattr(q, "_calib_spec") else None
                if spec is None or not w.is_cuda or w.dtype not in _lib._DTYPES or w.dtype == torch.float64 or w.numel() == 0:
                    continue
                wc = w.detach().contiguous()
                groups.setdefault((w.device, w.dtype, spec["stat"], spec["ovp"]), []).append((q, spec, wc, _tensor_key(w)))
            for (_, _, stat, ovp), items in groups.items():
                if len(items) < 2 or not (self.batch_calibration == 2 or any(len(spec["grids"]) > 1 for _, spec, _, _ in items)):
                    continue
                # (issued in a few calls, a small one first: the stream starts on it while the host prepares the rest).  Every
                # call's type picks travel to pinned memory behind it, with an event of their own: a quantiser waits for ITS
                # call only when its forward comes (round 5: the first layers' picks arrive after the first, small call; the
                # later calls finish while the host walks the first layers -- one 9-13 ms wait became ~2 ms)
                for b0, b1 in [(0, 4)] + [(b, b + 24) for b in range(4, len(items), 24)]:
                    jobs = []
                    for q, spec, wc, _ in items[b0:b1]:
                        rows = wc.shape[0]
                        plans = [_lib.plan_for(g) for g in spec["grids"]]
                        jobs.append((wc, rows, wc.numel() // rows, True, plans, spec["gmaxs"], spec["lb"], spec["ub"], spec["step"]))
                    if not jobs:
                        continue
                    r, t = _lib.calibrate_batch(jobs, xmax=stat, ovp=ovp)
                    host = torch.empty(len(jobs), dtype=torch.int32).pin_memory()
                    host.copy_(t, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(t.device))
                    for k, ((q, spec, wc, key), (alpha, score, _)) in enumerate(zip(items[b0:b1], r)):
                        q._calib_ready = (key, spec, _LazyPick(ev, host, k), alpha, score, wc.shape[0])
                self.precalibrated += len(items)

    def __reduce__(self):
        return (type(None), ())

    def disable(self):
        self.flush()
        self.enabled = False
        if self.bank is not None:
            self.bank.detach()
            self.bank = None

    def poke(self, q):
        """From tensor_forward of a calibrated weight quantiser that has no bank, outside autograd."""
        if not self.enabled or self.bank is not None or self.first is None or self.first() is not q or self.failed >= 3:
            return
        model = self._model()
        if model is None:
            return
        # One resident quantised copy of every weight: affordable for the models the reference evaluates, not for a model that
        # fills the device.  Build only when the copies fit comfortably into what is free right now; otherwise (and after an
        # allocation failure) the bank stays off for good and every layer keeps the reference's schedule.
        need = {}
        for _, _, q, w in _weight_layers(model):
            if w.is_cuda and q._steady:
                need[w.device] = need.get(w.device, 0) + w.numel() * w.element_size()
        for dev, n in need.items():
            free, _ = torch.cuda.mem_get_info(dev)
            free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)      # (cached blocks torch can reuse)
            if n > self.mem_fraction * free:
                self.enabled = False
                self.reason = "resident copies need %d MB, %d MB free on %s" % (n >> 20, free >> 20, dev)
                return
        try:
            bank = WeightBank(model, resident=self.resident)
        except _lib.AntqError:
            self.failed += 1
            return
        except (torch.cuda.OutOfMemoryError, RuntimeError) as ex:
            self.enabled = False
            self.reason = "building the bank failed: %s" % (str(ex).splitlines() or [""])[0]
            return
        if any(reason == "not calibrated yet" for _, reason in bank.skipped):
            bank.detach()              # (a partially calibrated model: try again on a later forward)
            return
        bank.auto = weakref.ref(self)
        self.bank = bank


class FlushHook:
    """Forward hook of the model enable_quantization armed, run when its forward returns: AutoBank.flush() (deferred type
    picks get their log lines), and -- default schedule -- the bank's copies are spent: the next forward re-quantises.
    Stateless (the AutoBank is looked up on the module), so a deep copy / an unpickled copy of an armed model carries a
    working hook; such a copy has lost its AutoBank (never copied or pickled) and gets a fresh one here, after its first
    forward."""

    def __init__(self, auto=None):
        pass

    def __deepcopy__(self, memo):
        return self

    def __call__(self, module, inputs, output):
        auto = module.__dict__.get("_antq_auto_bank")
        if auto is None:
            try:
                auto = AutoBank(module)
                object.__setattr__(module, "_antq_auto_bank", auto)
            except Exception:          # noqa: BLE001  (exotic containers: the per-layer schedule simply stays)
                return
        if auto.queue:
            auto.flush()
        if auto.bank is not None and not auto.bank.resident:
            auto.bank.dirty = True
        from . import core
        if core.search_memo.entries:
            core.search_memo.clear()       # (shared activations are shared within ONE forward: nothing outlives it)
