"""Host mirrors of a quantiser's tiny device buffers (`bit`, `has_inited_quant_para`).

The reference reads these buffers on every forward (`if self.has_inited_quant_para == 0`, `self.bit > 6`:
AQ:470, :482), which costs a device->host sync each time.  Here the host keeps the last value it read or wrote,
validated by (data_ptr, _version, device) of the buffer: an in-place edit (`fill_`, `copy_`, load_state_dict), a
rebinding (`.bit.data = torch.tensor(8)`, AQ/quant_model.py:83) or nothing at all are told apart without touching
the device; only a changed buffer is read back, once.  Device / dtype moves (`.to()`, `.cuda()`, `.half()`) keep
the value and re-key the mirror.

What the key cannot see: a write through `.data` that lands on the SAME address with the same `_version` (an in-place
`buf.data.copy_(...)`, or two rebindings between forwards whose second allocation recycles the first address).  The
in-package writers (`_to_8bit`, `set_8_bit_layer_*`, `load_state_dict`, calibration) re-key or re-arm themselves; external
code that edits `bit` / `has_inited_quant_para` / `quant_grid` that way must call the quantiser's `rearm()` afterwards
(INTEGRATION.md, section 2).  The operator-level drop-in (`quant_cuda.quant`) does not rely on keys at all: its kernel
verifies the grid on the device.
"""
import threading
import weakref

import torch


class HostMirrorMixin:
    # Host-side state of a Quantizer that is never a Parameter, a buffer or a submodule.  torch.nn.Module.__setattr__ spends
    # ~2.5 us per assignment on type checks and registry look-ups -- ~2000 assignments in BERT-base's calibrating forward,
    # 5 ms of a 38 ms host-bound call (profiles/r06_first_forward_host.log) -- so plain values of these names go straight into
    # the instance dictionary (what Module.__setattr__ ends up doing for them anyway).
    _PLAIN = frozenset((
        "mode", "name", "is_signed", "is_perchannel", "is_enable", "is_enable_activation", "is_enable_weight", "weights_at_rest",
        "_steady", "_plan", "_gmax", "_grid_key", "_searched", "_type_search", "_pending", "_calib_ready", "_calib_ctx",
        "_sign_probe", "_defer_allowed", "_grad_call", "_hm", "_rest_stamp", "_rest_src", "_alpha32_stamp", "_bank", "_auto_bank",
        "_spec_out", "_rest_out", "_alpha32"))

    def __setattr__(self, name, value):
        if name in HostMirrorMixin._PLAIN and not isinstance(value, (torch.Tensor, torch.nn.Module)):
            self.__dict__[name] = value
        else:
            super().__setattr__(name, value)

    def _hm_setup(self, **values):
        self._hm = {name: [self._hm_key(name), value] for name, value in values.items()}

    def _hm_key(self, name):
        t = self._buffers.get(name)                  # (the mirrored values are buffers: past Module.__getattr__'s miss-then-search)
        if t is None:
            t = getattr(self, name)
        return (t.data_ptr(), t._version, t.device)

    def _hm_get(self, name):
        ent = self._hm[name]
        key = self._hm_key(name)
        if ent[0] != key:
            ent[1] = getattr(self, name).item()      # the buffer changed behind our back: one read-back
            ent[0] = key
        return ent[1]

    def _hm_known(self, name, value):
        """The host just wrote `value` into the buffer itself."""
        self._hm[name] = [self._hm_key(name), value]

    def _hm_fresh(self):
        return all(ent[0] == self._hm_key(name) for name, ent in self._hm.items())

    def _apply(self, fn, *args, **kwargs):
        fresh = {name: ent[0] == self._hm_key(name) for name, ent in self._hm.items()}
        out = super()._apply(fn, *args, **kwargs)
        for name, ok in fresh.items():
            if ok:
                self._hm[name][0] = self._hm_key(name)
        return out


def _lib_tensor_key(t):
    """What has to be unchanged for a search run ahead of the forward to stand for the tensor now at hand."""
    if not isinstance(t, torch.Tensor) or torch.is_inference(t):
        return None
    return (t.data_ptr(), t._version, t.dtype, tuple(t.shape), t.is_contiguous())


class _PinnedSlots:
    """Pinned host floats for device->host copies nobody wants to wait for at issue time.  A slot belongs to whoever took it
    until it is given back (`give`): a type pick parked until the end of the model's forward is never overwritten by a
    later layer's probe, however many layers the model has (GPT-2 XL / OPT: 192 linears, two slots each).  The pool grows
    by one pinned chunk whenever the free list runs dry.  Process-global and shared by threads (nn.DataParallel replicas,
    multi-threaded serving): take / give hold a lock.  A slot taken on behalf of an `owner` (a quantiser) returns by itself
    when the owner is collected without having given it back -- a model deleted half way through a calibrating forward, an
    exception between take() and the read -- so nothing leaks."""

    def __init__(self, chunk=256, alloc=None):
        self.chunk, self.free, self.chunks = chunk, [], []
        self._alloc = alloc or (lambda n: torch.zeros(n, dtype=torch.float32).pin_memory())
        self._lock = threading.Lock()
        self._out = {}             # id(slot) -> (slot, finalizer or None): slots currently taken

    def take(self, owner=None):
        with self._lock:
            if not self.free:
                buf = self._alloc(self.chunk)
                self.chunks.append(buf)
                self.free.extend(buf[i:i + 1] for i in range(self.chunk - 1, -1, -1))
            slot = self.free.pop()
            fin = weakref.finalize(owner, self._reclaim, id(slot)) if owner is not None else None
            if fin is not None:
                fin.atexit = False
            self._out[id(slot)] = (slot, fin)
            return slot

    def give(self, slot):
        if slot is None:
            return
        with self._lock:
            ent = self._out.pop(id(slot), None)
            if ent is None:
                return                              # (already back: reclaimed, or given twice)
            if ent[1] is not None:
                ent[1].detach()
            self.free.append(ent[0])

    def _reclaim(self, key):
        with self._lock:
            ent = self._out.pop(key, None)
            if ent is not None:
                self.free.append(ent[0])

    def outstanding(self):
        with self._lock:
            return len(self._out)


_slots = _PinnedSlots()


class CalibrationMixin:
    """First-call calibration without stalling the stream (shared by the ANT and the OliVe Quantizer).

    * The sign probe.  An input quantiser starts unsigned and learns from its first tensor whether it needs a signed codebook
      (`if tensor.min() < 0`, AQ:61-63 / OQ:62-64): a device->host read in front of everything else its calibration does.
      Read on the spot it drains the stream once per layer -- and the GPU then idles while the host issues the calibration
      launches.  The wrapper layers therefore call `prefetch_sign(input)` BEFORE they calibrate their weight: the minimum goes
      to pinned memory asynchronously, the weight's calibration is issued behind it, and when the input quantiser asks, the
      answer has long arrived while the stream is still busy.  Without a matching probe `update_signed` reads on the spot.
    * `mse`, the log value of the reference (AQ:519-520), is formed when somebody reads it.
    """
    _sign_probe = None
    _calib_ready = None           # (weight key, spec, type pick (lazy), alpha [types, rows], score [types], rows) from weight_bank.AutoBank.precalibrate
    _calib_ctx = None             # the model's weight_bank.AutoBank (weight AND input quantisers): log order, deferred picks
    _pending = None               # (event, pinned slot, spec): a type pick made on the device, not yet known to the host
    _spec_out = None              # the calibrating forward's output of a quantiser whose pick is pending
    _defer_allowed = False        # set by tensor_forward: no gradient is wanted through this call
    _grad_call = False            # set by tensor_forward: the caller's forward wants gradients for this tensor or alpha

    def _before_calibration(self, tensor):
        """First thing in tensor_forward of an enabled quantiser.  A weight quantiser that is not calibrated yet lets the model
        search all its weights at once (weight_bank.AutoBank.precalibrate); one whose search is done installs the result --
        provided the tensor in its hands is still the one that was searched (address, version, shape)."""
        if self._pending is not None:                 # (a second forward before the model-level flush: learn the pick now)
            if self._calib_ctx is not None:
                self._calib_ctx.flush()
            if self._pending is not None:
                self._emit(self._resolve_pending())
        if self._steady or self.is_input:
            return
        if self._calib_ready is None and self._auto_bank is not None:
            self._auto_bank.precalibrate()
        ready, self._calib_ready = self._calib_ready, None
        if ready is not None and ready[0] == _lib_tensor_key(tensor) and self._hm_get('has_inited_quant_para') == 0:
            with torch.no_grad():
                key, spec, pick, alpha, score, rows = ready
                t = pick.get() if hasattr(pick, "get") else int(pick)       # (waits for the pick's own batch call only)
                self._calib_apply(spec, t, alpha[t], score[t:t + 1], rows)

    def _emit(self, line):
        """The reference's calibration log line (`print(mode, end="\\t"); print("%d-bit \\t %s," ...)`), kept in the order the
        quantisers calibrate even when an earlier quantiser's line still waits for its type pick."""
        if self._calib_ctx is not None:
            self._calib_ctx.emit(line)
        elif line:
            print(line)

    # ---- type selection without a stream drain (input quantisers of `ant-...` modes) -------------------------------------
    # The host needs the picked type only to NAME it (mode string, host plan); the data path does not: the clip search of
    # every candidate codebook and the pick run in one stream-ordered call (antq_calibrate), alpha and the codebook buffers
    # are gathered by the device-side index, and this forward's output is the fake-quantised tensor of EVERY candidate with
    # the picked one gathered the same way (T short launches and one gather instead of a drain: 73 drains per BERT-base
    # forward).  The pick travels to pinned memory behind all that; the model's forward hook (or the quantiser's next
    # forward) reads it, names the mode, builds the host plan and prints the line.
    def _defer_ok(self, data):
        ctx = self._calib_ctx
        if ctx is None or not ctx.defer_types or not self.is_input or self.is_perchannel or not self._defer_allowed:
            return False
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            return False
        return data.is_cuda and data.dtype in (torch.float32, torch.bfloat16, torch.float16) and data.numel() > 0

    def _calibrate_deferred(self, data, spec):
        from . import _lib, core
        x = data.detach().contiguous()
        plans = [_lib.plan_for(g) for g in spec["grids"]]
        key = ("calibrate", spec["stat"], spec["ovp"], spec["lb"], spec["ub"], spec["step"],
               tuple(g.tobytes() for g in spec["grids"]), tuple(spec["gmaxs"]))
        seen = core.search_memo.get(data, key)
        if seen is None:
            alpha, score, typ, _ = _lib.calibrate(x, 1, x.numel(), False, plans, spec["gmaxs"], spec["lb"], spec["ub"],
                                                  spec["step"], xmax=spec["stat"], ovp=spec["ovp"])
            core.search_memo.put(data, key, (alpha, score, typ))
        else:
            alpha, score, typ = seen
        # the installed state and this call's output from the device-side pick: two launches (antq_calibrate_install, round 6)
        # -- alpha, mse and quant_grid gathered by the pick, ONE pass over the tensor with the picked codebook -- instead of a
        # dozen small torch launches, a fake-quant pass per candidate and a gather of the whole tensor
        rows_np = self._selectable_rows(spec)
        installed = False
        if rows_np is not None and x.dim() >= 1:
            stack = core.device_grid_master(rows_np, x.device)
            # (alpha and the codebook in storages of their own: an in-place edit of one must not move the other's version counter)
            am = torch.empty(2, dtype=torch.float32, device=x.device)                  # [alpha, mse]
            grid_out = torch.empty(stack.shape[1], dtype=torch.float32, device=x.device)
            out = torch.empty_like(x)
            installed = _lib.calibrate_install(x, out, plans, spec["gmaxs"], typ, alpha.reshape(-1), score.reshape(-1), stack,
                                               grid_out, am[0:1], am[1:2], ovp=spec["ovp"])
            if installed:
                self.alpha.data = am[0].reshape(())
                self.quant_grid.data = grid_out
                self._after_install_selected(spec)
                self._searched = True
                self.mse = am[1]
                self._spec_out = out.view(data.shape)
        if not installed:
            idx = typ.long()
            self.alpha.data = alpha.index_select(0, idx).reshape(())
            self._install_selected(spec, idx)
            self._searched = True
            self._mse_later(score.index_select(0, idx), 1)
            outs = torch.empty((len(plans),) + tuple(x.shape), dtype=x.dtype, device=x.device)
            for t, (p, gm) in enumerate(zip(plans, spec["gmaxs"])):
                core.fake_quant(x, alpha[t], p, gm, False, ovp=spec["ovp"], out=outs[t])
            self._spec_out = outs.index_select(0, idx)[0].view(data.shape)
        slot = _slots.take(self)
        slot.copy_(typ.float(), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(data.device))
        self._pending = (ev, slot, spec)
        self.has_inited_quant_para.data = torch.ones_like(self.has_inited_quant_para)
        self._hm_known('has_inited_quant_para', 1.0)
        self._calib_ctx.defer(self)

    def _resolve_pending(self):
        """The pick has arrived (or is waited for): name the mode, adopt the host plan; returns the log line."""
        ev, slot, spec = self._pending
        self._pending = None
        ev.synchronize()
        t = int(slot[0])
        _slots.give(slot)
        if not 0 <= t < len(spec["modes"]):
            raise RuntimeError("type pick %r out of range for %s" % (t, spec["modes"]))
        self.mode = spec["modes"][t]
        self._adopt_plan(spec, t)
        self._steady = True
        return self._calib_line()

    def prefetch_sign(self, tensor):
        if self.is_signed or self._steady or not isinstance(tensor, torch.Tensor) or not tensor.is_cuda or tensor.numel() == 0:
            return
        if not self.is_enable or self.mode == "base" or not (self.is_enable_activation if self.is_input else self.is_enable_weight):
            return
        if self._hm_get('has_inited_quant_para') != 0:
            return
        self._drop_probe()
        with torch.no_grad():
            slot = _slots.take(self)
            slot.copy_(tensor.detach().min().float().reshape(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(tensor.device))
        self._sign_probe = (tensor, _tensor_stamp(tensor), slot, ev)

    def _drop_probe(self):
        probe, self._sign_probe = self._sign_probe, None
        if probe is not None:
            probe[3].synchronize()                       # (the copy into the slot has landed: it may be reused)
            _slots.give(probe[2])

    def update_signed(self, tensor):
        if self.is_signed:                               # already signed (weights): nothing to learn, no sync
            return
        probe, self._sign_probe = self._sign_probe, None
        if probe is not None:
            probe[3].synchronize()                       # (recorded before the weight's calibration was issued)
            value = float(probe[2][0])
            _slots.give(probe[2])
        if probe is not None and probe[0] is tensor and probe[1] is not None and probe[1] == _tensor_stamp(tensor):
            negative = value < 0
        else:
            negative = bool(tensor.min() < 0)
        if negative:
            self.is_signed = True

    @property
    def mse(self):
        lazy = self.__dict__.get('_mse_lazy')
        if lazy is not None:
            best, n = lazy
            self.__dict__['_mse_val'] = best.sum() / n
            self.__dict__['_mse_lazy'] = None
        return self.__dict__.get('_mse_val')

    @mse.setter
    def mse(self, value):
        self.__dict__['_mse_val'] = value
        self.__dict__['_mse_lazy'] = None

    def _mse_later(self, best_score, n):
        self.__dict__['_mse_lazy'] = (best_score, n)


class WeightsAtRestMixin:
    """The weights-at-rest launch mode of a WEIGHT quantiser (quant_utils.set_weights_at_rest), shared by the ANT and the
    OliVe Quantizer: the steady-state launch may start while earlier work on the stream drains (ANTQ_FLAG_UNORDERED) when
    its inputs are provably not being written by anything in flight.  Needs: weights_at_rest, is_input, alpha, _plan, _gmax
    and the four _rest_* / _alpha32* slots set to None in the constructor."""

    def _rest_on(self):
        return self.weights_at_rest and not self.is_input

    def _weights_may_change(self):
        """A training forward of this quantiser, or a train() / eval() call on it: an optimiser step may follow (or have
        happened), and optimisers that write through `.data` (the reference's BertAdam, BERT/optimization.py:161) move no
        version counter.  Everything keyed on (address, version) is dropped: the attached WeightBank re-quantises on its next
        lookup, the weights-at-rest mode launches ordered once and re-reads alpha."""
        bank = self.__dict__.get("_bank")
        if bank is not None:
            bank.dirty = True
        self._rest_stamp = self._rest_src = None
        self._alpha32_stamp = None

    def train(self, mode=True):
        self._weights_may_change()
        return super().train(mode)

    def __getstate__(self):
        """torch.save(model) pickles the module: the launch-mode caches (a weak reference among them) are not state."""
        st = self.__dict__.copy()
        for k in ("_rest_src", "_rest_stamp", "_rest_out", "_alpha32", "_alpha32_stamp", "_sign_probe", "_pending", "_spec_out",
                  "_calib_ctx", "_calib_ready", "_bank", "_auto_bank"):
            if k in st:
                st[k] = None
        return st

    def _rest_buffer(self, data):
        """The quantiser-owned output buffer of the weights-at-rest mode (None otherwise)."""
        if not self._rest_on() or self._grad_call or torch.is_grad_enabled() and data.requires_grad:
            return None
        b = self._rest_out
        if b is None or b.shape != data.shape or b.dtype != data.dtype or b.device != data.device:
            with torch.inference_mode(False):      # an ordinary tensor even when the forward runs under inference_mode
                b = self._rest_out = torch.empty_like(data, memory_format=torch.contiguous_format)
            torch.cuda.current_stream(data.device).synchronize()      # (once: nothing in flight may still own this block)
        return b

    def _rest_alpha(self):
        """alpha as the kernels take it (float32).  A model moved to bf16 / fp16 carries a 16-bit alpha Parameter; converting
        it on every forward would put a kernel in flight right in front of the launch (which then has to stay ordered), so
        the weights-at-rest mode keeps a float32 copy, refreshed when the Parameter's storage or version changes."""
        a = self.alpha
        if a.dtype == torch.float32 or not self._rest_on() or self._grad_call or (torch.is_grad_enabled() and a.requires_grad):
            return a
        st = _tensor_stamp(a)
        if st is None:                             # an inference tensor: no version counter, nothing to key a cache on
            return a
        if self._alpha32_stamp != st:
            with torch.inference_mode(False):      # an ordinary tensor even when the forward runs under inference_mode
                self._alpha32 = a.detach().to(torch.float32).reshape(-1).contiguous()
            self._alpha32_stamp = st
        return self._alpha32

    def _at_rest(self, data):
        """Whether THIS call may launch unordered: the weight is the very tensor OBJECT the previous call saw (held by a weak
        reference: a temporary -- w.t().contiguous(), a slice copy, a dequantised weight -- is a new object every forward even
        when the caching allocator hands it a recycled address, and never qualifies), with the same storage and version
        counter; alpha likewise; and the codebook (plan, gmax) is the one the previous call used, so nothing may still be
        writing or re-planning what this launch reads.  Whatever changed them (calibration a moment ago, load_state_dict, an
        optimiser step, a dtype / device move, a new grid) is then at least one ordinary, ordered launch of this quantiser in
        the past.  The first call after any such change launches ordered; so does every call on a tensor without a version
        counter (created under torch.inference_mode), every training forward, and the first call after a training forward or
        a train() / eval() call (`_weights_may_change`: optimisers that write through `.data`).  (A `.data` edit between two
        no-grad forwards with none of those in between stays the caller's promise.)"""
        if not self._rest_on():
            return False
        if self._grad_call or (torch.is_grad_enabled() and (data.requires_grad or self.alpha.requires_grad)):
            self._weights_may_change()             # (a training forward: the step that follows may write through `.data`)
            return False
        ds, as_ = _tensor_stamp(data), _tensor_stamp(self.alpha)
        if ds is None or as_ is None:
            self._rest_stamp = self._rest_src = None
            return False
        stamp = (ds, as_, id(self._plan), self._gmax)
        src = self._rest_src() if self._rest_src is not None else None
        if src is not data or stamp != self._rest_stamp:
            self._rest_stamp = stamp
            self._rest_src = weakref.ref(data)
            return False
        return True


def _tensor_stamp(t):
    """(storage address, version counter) of a tensor, None for an inference tensor (torch raises on its _version)."""
    if torch.is_inference(t):
        return None
    return (t.data_ptr(), t._version)
