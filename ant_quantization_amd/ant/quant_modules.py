"""MI355X-native mirror of ant_quantization/antquant/quant_modules.py (ANT, MICRO'22).

Same classes, constructor arguments, buffers / state-dict keys, mode strings and printed
lines as the reference, so `quantize_model`, checkpoints and the ImageNet / BERT harnesses
work unchanged -- but the arithmetic runs in hand-written gfx950 kernels:

  reference (per forward)                          here
  -----------------------------------------------  -----------------------------------------
  x/scale, quant_cuda.quant, (q-d)+d, *scale       ONE fused kernel (antq_fakequant)
  = 7 PyTorch/CUDA launches, 56 B/elem              8 B/elem fp32, 4 B/elem bf16
  search_mse: 75 x (_forward + mse_loss)            ONE kernel evaluates all candidates on a
  = ~750 launches per grid type                     single read of the tensor (antq_search_sse)
  `if cuda_tensor:` host syncs every call           host-side mirrors of bit / inited / signed

There is no CPU path: tensors must live on a HIP device (the reference's quantiser is
CUDA-only as well; its only CPU-capable piece is quant_affine.py).
"""
import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, core, grids
from .._mirror import CalibrationMixin, HostMirrorMixin, WeightsAtRestMixin
from .quant_affine import *  # noqa: F401,F403  (the reference star-imports it, AQ:9)

_TYPE_ORDER = ("int", "flint", "pot", "float", "float1", "float2", "float3", "float4", "apot")


def _dist_on():
    return dist.is_available() and dist.is_initialized()


def _rank():
    return dist.get_rank() if _dist_on() else 0


class QuantBase():
    """AQ:11-24.  `_quantization` is the operator boundary: flat view, grid cast to x's
    dtype, nearest-value kernel, reshape."""

    def _quantization(x, quant_grid, plan=None):
        """The operator boundary (AQ:12-18 / OQ:9-15): flat view, the nearest-value operator, reshape back -- what
        `quant_cuda.quant` does, minus the all-zero index tensor the reference allocates and throws away.  `plan`: the
        calling quantiser's own plan of `quant_grid`, passed to the kernel as a hint it verifies against the buffer."""
        flat = x.view(-1).contiguous()
        if plan is not None and _lib.hinted_ok(flat, quant_grid, plan):
            z = _lib.nearest_hinted(flat, quant_grid.contiguous(), plan)
        else:
            g = quant_grid.type_as(flat) if flat.dtype in (torch.float32, torch.float64) else quant_grid.float()
            z = _lib.nearest(flat, g.contiguous())
        return z.view(x.shape)

    @staticmethod
    def forward(real_val, quant_grid, plan=None):
        with torch.no_grad():
            return QuantBase._quantization(real_val, quant_grid, plan)


class Quantizer(HostMirrorMixin, WeightsAtRestMixin, CalibrationMixin, nn.Module):
    def __init__(self, mode="base", bit=8, is_signed=True, is_enable=False, is_input=False, args=None, operator=None):
        super(Quantizer, self).__init__()
        self.mode = mode
        self.is_input = is_input
        self.is_signed = is_signed
        self.is_enable = is_enable
        self.is_enable_activation = is_enable
        self.is_enable_weight = is_enable
        self.args = args
        self.operator = operator

        # same registration order / names as AQ:39-42 (state-dict compatibility)
        self.alpha = nn.Parameter(torch.tensor(1.0, requires_grad=True))
        self.register_buffer('bit', torch.tensor(bit))
        self.register_buffer('has_inited_quant_para', torch.tensor(0.0))
        self.register_buffer('quant_grid', torch.ones(2 ** bit))
        self._hm_setup(bit=int(bit), has_inited_quant_para=0.0)

        self.w_up = self.args.w_up
        self.a_up = self.args.a_up
        self.w_low = self.args.w_low
        self.a_low = self.args.a_low

        self.percent = self.args.percent / 100
        self.is_perchannel = True
        if is_input:
            # Input shouldn't be per-channel quantizaton
            self.is_perchannel = False
        self.search = args.search
        self.mse = torch.tensor(0.0)

        ## debug
        self.name = None

        # host-side knowledge (never read back from the device in steady state)
        self._steady = False      # calibrated and nothing re-armed since
        self._plan = None         # _lib.Plan of the installed grid
        self._gmax = 10.0
        self._grid_key = None
        self._searched = False    # the last search_mse had at least one candidate
        self._bank = None         # weight_bank.WeightBank serving this quantiser's output, if attached
        self._auto_bank = None    # weight_bank.AutoBank armed by enable_quantization(model)
        self.weights_at_rest = False   # opt-in (quant_utils.set_weights_at_rest): this WEIGHT quantiser's tensor and alpha are
                                       # not written by anything still in flight when forward runs (inference on frozen
                                       # weights), so its launch may start while earlier work on the stream drains.  The
                                       # output then lives in ONE buffer owned by the quantiser (a fresh tensor from the
                                       # stream-ordered allocator could sit on memory an in-flight kernel still uses):
                                       # every forward returns the same storage, like WeightBank's resident outputs
        self._rest_out = None
        self._rest_stamp = None
        self._rest_src = None
        self._alpha32 = None
        self._alpha32_stamp = None
        self._type_search = None  # during one calibration: grid bytes -> clip search result of the type selection's pass

    # ---------------------------------------------------------------- bookkeeping
    def disable_input_quantization(self):
        self.is_enable_activation = False

    def enable_quantization(self, name):
        self.name = name
        self.is_enable = True

    def disable_quantization(self, name):
        self.name = name
        self.is_enable = False

    # (update_signed / prefetch_sign / the lazily formed `mse`: _mirror.CalibrationMixin)

    def _load_from_state_dict(self, state_dict, prefix, *a, **k):
        super()._load_from_state_dict(state_dict, prefix, *a, **k)
        self._steady = False       # buffers may have changed: re-read them once
        self._plan = None

    def rearm(self):
        """Force the next forward to re-read `bit` / `has_inited_quant_para` / `quant_grid` from the device, whatever the
        buffers' (data_ptr, _version) keys say (set_8_bit_layer_*, and any outside edit through `.data`)."""
        self._steady = False
        self._plan = None
        for ent in self._hm.values():
            ent[0] = None

    # ---------------------------------------------------------------- grids (AQ:75-278)
    def _bits(self):
        return int(self._hm_get('bit'))

    def _to_grid(self, values):
        return core.device_grid(values, self.quant_grid.device)

    def int_value(self, q_type="int"):
        return self._to_grid(grids.ant_int(self._bits(), self.is_signed))

    def flint_value(self, exp_base=0):
        return self._to_grid(grids.ant_flint(self._bits(), self.is_signed))

    def pot_value(self):
        return self._to_grid(grids.ant_pot(self._bits(), self.is_signed))

    def float_value(self, eb=3):
        return self._to_grid(grids.ant_float(self._bits(), self.is_signed, eb))

    def apot_value(self):
        return self._to_grid(grids.ant_apot(self._bits(), self.is_signed))

    def _install_grid(self, values):
        """quant_grid <- values (host numpy), plan built on the host: no device read-back."""
        v = np.ascontiguousarray(values, dtype=np.float32)
        self.quant_grid.data = self._to_grid(v)
        self._plan = _lib.plan_for(v)
        with np.errstate(all="ignore"):
            self._gmax = float(np.max(v))
        self._grid_key = self._grid_now()

    def _grid_now(self):
        g = self.quant_grid
        return (g.data_ptr(), g._version)

    def _ensure_plan(self):
        """The plan of the grid currently in `quant_grid`.  The buffer is watched by (data_ptr, _version): a checkpoint
        load, the DDP broadcast of AQ:531 or a user's edit replaces the plan (one read-back), nothing else costs a sync."""
        if self._plan is None or self._grid_key != self._grid_now():
            v = self.quant_grid.detach().float().cpu().numpy()
            self._plan = _lib.plan_for(v)
            self._gmax = float(np.max(v))
            self._grid_key = self._grid_now()
        return self._plan

    # ---------------------------------------------------------------- calibration
    def mse_loss(self, quant_tensor, source_tensor, p=2.0, is_perchannel=True):
        if is_perchannel:
            mean_tensor = (quant_tensor - source_tensor).abs().pow(p).view(quant_tensor.shape[0], -1).mean(-1).unsqueeze(1)
            return mean_tensor
        else:
            return (quant_tensor - source_tensor).abs().pow(p).mean()

    def _search_window(self, per_channel):
        """[lb, ub) of the clip search in percent of x_max (AQ:293-296 / :311-314): the configured window, its lower end
        raised to 95 for layers of more than 6 bits.  One helper for search_mse and the fused type selection."""
        lb = int(self.w_low) if per_channel else int(self.a_low)
        ub = int(self.w_up) if per_channel else int(self.a_up)
        if self._bits() > 6:
            lb = int(95)
        return lb, ub

    def _clip_search(self, tensor, need_xmax=True):
        """search_mse's device work: (best_score [rows or 1], alpha [rows or 1], x_max), nothing reduced or reshaped yet."""
        per_channel = self.is_perchannel and (not self.is_input)
        x_max = None
        lb, ub = self._search_window(per_channel)
        plan = self._ensure_plan()
        # (keyed on the window as well: a pass over another [lb, ub) can never stand in for this search)
        hit = self._type_search.get((plan.grid.tobytes(), lb, ub)) if self._type_search else None
        if hit is not None:
            best_score, alpha, ratios = hit       # this very search was part of the type selection's single pass
        else:
            # (a parallel branch's input quantiser may have searched this very tensor a moment ago: core.SearchMemo)
            key = ("ant", lb, ub, 1, plan.grid.tobytes(), self._gmax) if not per_channel else None
            seen = core.search_memo.get(tensor, key) if key else None
            if seen is not None:
                best_score, alpha, ratios = seen[0], seen[1].clone(), seen[2]
            else:
                x_max = core.row_absmax(tensor, per_channel)
                best_score, alpha, ratios = core.clip_search(tensor, x_max, per_channel, lb, ub, 1, plan, self._gmax)
                if key:
                    core.search_memo.put(tensor, key, (best_score, alpha.clone(), ratios))
        self._searched = ratios is not None
        if x_max is None and need_xmax:
            x_max = core.row_absmax(tensor, per_channel)
        return best_score, alpha, x_max, per_channel

    def search_mse(self, tensor):
        """AQ:287-326: clip search over i in [lb, ub), alpha_i = x_max * (i*0.01)."""
        best_score, alpha, x_max, per_channel = self._clip_search(tensor)
        ratio = (alpha / x_max).mean()        # 0-dim tensor: the reference's float, without the sync
        if per_channel:
            return best_score.sum(), alpha.unsqueeze(1), ratio
        return best_score.sum(), alpha.reshape(()), ratio

    def search_adaptive_numeric_type(self, data):
        """AQ:328-415: per-tensor choice of the type with the smallest summed best-MSE.
        Quirk kept: the -float1..4 searches all use float_value(1) (AQ:370-397)."""
        modes, mse_list, type_grids = [], [], []
        mode = self.mode
        bit, signed = self._bits(), self.is_signed
        for t in _TYPE_ORDER:
            if ("-" + t) not in mode:
                continue
            if t.startswith("float") and t != "float":
                g = grids.ant_float(bit, signed, 1)
            else:
                g = grids.ant_grid(t, bit, signed)
            modes.append(t)
            type_grids.append(np.ascontiguousarray(g, dtype=np.float32))
        # every candidate type's clip search on ONE read of the tensor (antq_search_sse_multi); the search on the grid
        # that is installed afterwards is one of them and is not repeated (search_mse looks it up)
        per_channel = self.is_perchannel and (not self.is_input)
        lb, ub = self._search_window(per_channel)
        uniq = list({g.tobytes(): g for g in type_grids}.values())          # (-float1..4 all search float_value(1))
        plans = [_lib.plan_for(g) for g in uniq]
        with np.errstate(all="ignore"):
            gmaxs = [float(np.max(g)) for g in uniq]
        x_max = core.row_absmax(data, per_channel)
        key = ("ant-types", lb, ub, 1, tuple(g.tobytes() for g in uniq), tuple(gmaxs)) if not per_channel and len(uniq) > 1 else None
        res = core.search_memo.get(data, key) if key else None
        if res is not None:
            res = [(b, a.clone(), r) for b, a, r in res]
        else:
            res = core.clip_search_types(data, x_max, per_channel, lb, ub, 1, plans, gmaxs) if len(uniq) > 1 else None
            if key and res is not None:
                core.search_memo.put(data, key, [(b, a.clone(), r) for b, a, r in res])
        if res is not None:
            self._type_search = {(g.tobytes(), lb, ub): r for g, r in zip(uniq, res)}
            mse_list = [self._type_search[(g.tobytes(), lb, ub)][0].sum().reshape(()) for g in type_grids]
        else:
            for t, g in zip(modes, type_grids):
                self.mode = t
                self._install_grid(g)
                best, _, _ = self.search_mse(data)
                mse_list.append(best.reshape(()))
        mse_idx = np.argsort(torch.stack(mse_list).cpu().numpy())     # one read-back for all types
        self.mode = modes[mse_idx[0]]

    def outlier_set(self, data):
        """'outlier' baseline mode (AQ:417-436): an int4 body up to the `percent`-th percentile of |x| and an
        int16 grid for what lies beyond; the two range ends are averaged over DDP ranks."""
        absx = data.abs()
        p4 = torch.tensor(np.percentile(absx.cpu().numpy(), self.percent * 100), device=data.device)
        p16 = absx.max()
        if _dist_on():
            for t in (p4, p16):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                t /= dist.get_world_size()
        self.percent_value_int4, self.percent_value_int16 = p4, p16
        if _rank() == 0:
            print(self.name, p4.item(), p16.item())
        self.is_perchannel = False
        self._install_grid(grids.ant_int(self._bits(), self.is_signed))
        self.has_inited_quant_para.data = torch.ones_like(self.has_inited_quant_para)
        self._hm_known('has_inited_quant_para', 1.0)
        self._steady = True

    def outlier_quant(self, data):
        """AQ:438-465.  Body: nearest int4 value at scale p4/max(grid); elements with |x| > p4 are re-quantised
        on a uniform int16 grid over (p4, p16] with a straight-through combination.  Same fp32 op order as the
        reference, written with `where` instead of masked assignment."""
        p4, p16 = self.percent_value_int4, self.percent_value_int16
        if p4 > 0:
            scale = p4 / torch.max(self.quant_grid)
            body = QuantBase.forward(data / scale, self.quant_grid, self._ensure_plan()).clone().detach() * scale
        else:
            body = data.clone().detach()
        if not (self.percent < 100):
            return body
        level = 2 ** 16 - 1 if self.is_signed else 2 ** 15 - 1
        step = (p16 - p4) / level
        mag = ((data.abs() - p4) / step).round() * step + p4
        fine = mag * data.sign()
        return torch.where(data.abs() > p4, (fine - body).detach() + body, body)

    def _init_quant_para(self, data, data_b):
        """AQ:468-533.  The device read of `has_inited_quant_para` happens once; afterwards the
        host flag `_steady` short-circuits (the reference syncs on it every forward)."""
        if self._steady and self._hm_fresh():
            return
        with torch.no_grad():
            self._hm_get('bit')                      # (re-keys the mirror if the buffer was edited or rebound)
            if self._hm_get('has_inited_quant_para') != 0:
                self._ensure_plan()
                self._steady = True
                return
            self.update_signed(data)

            if "ant-" in self.mode and self._bits() <= 6 and self._defer_ok(data):
                spec = self._calib_spec(data)
                if spec is not None and len(spec["grids"]) > 1:
                    return self._calibrate_deferred(data, spec)    # (the type pick stays on the device)

            if self.is_perchannel:
                x_max = core.row_absmax(data, True)
                self.alpha.data = x_max.unsqueeze(1)
            else:
                self.alpha.data = core.row_absmax(data, False).reshape(())

            if self.mode == 'outlier':
                return self.outlier_set(data)

            if self._bits() > 6:
                self.mode = 'int'
            else:
                if "ant-" in self.mode:
                    self.search_adaptive_numeric_type(data)
                    if _dist_on():
                        # The reference picks `mode` per rank and then broadcasts rank 0's quant_grid (AQ:531): ranks
                        # whose near-tied types resolve differently end up with a mode string that contradicts their
                        # grid.  Agree on rank 0's pick instead (identical whenever the ranks agreed anyway).
                        pick = torch.tensor([_TYPE_ORDER.index(self.mode)], device=data.device)
                        dist.broadcast(pick, 0)
                        self.mode = _TYPE_ORDER[int(pick.item())]

            if self.mode not in _TYPE_ORDER:
                raise RuntimeError("Unsupported mode: " + self.mode)
            self._install_grid(grids.ant_grid(self.mode, self._bits(), self.is_signed))

            # (search_mse without its two log values -- the summed score and the mean clip ratio, AQ:326 -- which nothing
            #  here reads: four small launches per quantiser less on the calibration pass)
            best_score, alpha, _, per_channel = self._clip_search(data, need_xmax=False)
            self.alpha.data = alpha.unsqueeze(1) if per_channel else alpha.reshape(())

            # AQ:519-520 runs _forward + mse_loss once more for the log value `mse`; that MSE is the winning
            # candidate's score, which the search already holds (mean over rows of the per-row best): formed when read.
            if self._searched:
                self._mse_later(best_score, self.alpha.numel() if self.is_perchannel else 1)
            else:
                self.mse = self.mse_loss(self._forward(data), data, 2, is_perchannel=self.is_perchannel).mean()
            if _dist_on():
                dist.broadcast(self.mse, 0)
            self._emit(self._calib_line())
            if _dist_on():
                rt = self.alpha.data.clone()
                dist.all_reduce(rt, op=dist.ReduceOp.SUM)
                rt /= dist.get_world_size()
                self.alpha.data = rt
                dist.broadcast(self.quant_grid, 0)

            self.has_inited_quant_para.data = torch.ones_like(self.has_inited_quant_para)
            self._hm_known('has_inited_quant_para', 1.0)
            self._steady = True
            self._type_search = None
            core.forget_absmax()

    # ---------------------------------------------------------------- calibrated ahead of the forward (weight_bank.precalibrate)
    def _calib_spec(self, tensor):
        """What _init_quant_para would search for this quantiser (AQ:468-533): the candidate types in the order of
        search_adaptive_numeric_type, their codebooks (duplicates searched once, the first of equals wins as np.argsort's
        does), the window -- or None when the quantiser keeps the step-by-step path.  Weight quantisers (signed, per
        channel): weight_bank.AutoBank.precalibrate; input quantisers (per tensor, sign already learnt): the type pick on
        the device, _mirror.CalibrationMixin._calibrate_deferred."""
        per_channel = self.is_perchannel and (not self.is_input)
        if self.mode in ("base", "outlier") or (not self.is_input and not (per_channel and self.is_signed)):
            return None
        if self.is_input and (self.is_perchannel or not self.is_enable_activation):
            return None
        if not self.is_enable or (not self.is_input and not self.is_enable_weight):
            return None
        if self._steady or self._hm_get('has_inited_quant_para') != 0:
            return None
        bit, signed = self._bits(), self.is_signed
        if bit > 6:
            modes = ["int"]
        elif "ant-" in self.mode:
            modes = [t for t in _TYPE_ORDER if ("-" + t) in self.mode]
            if not modes or any(t in ("float1", "float2", "float3", "float4") for t in modes):
                return None                # (those search float_value(1) and install another grid, AQ:370-397)
        elif self.mode in _TYPE_ORDER:
            modes = [self.mode]
        else:
            return None                    # (the step-by-step path raises the reference's error)
        lb, ub = self._search_window(per_channel)
        if not range(lb, ub, 1):
            return None
        uniq = {}
        for t in modes:
            g = np.ascontiguousarray(grids.ant_grid(t, bit, signed), dtype=np.float32)
            uniq.setdefault(g.tobytes(), (t, g))
        with np.errstate(all="ignore"):
            return dict(modes=[t for t, _ in uniq.values()], grids=[g for _, g in uniq.values()],
                        gmaxs=[float(np.max(g)) for _, g in uniq.values()], lb=lb, ub=ub, step=1, stat="absmax", ovp=False)

    def _calib_line(self):
        return ("%s\t%d-bit \t %s," % (self.mode, self._bits(), self.name)) if _rank() == 0 else ""

    def _install_selected(self, spec, idx):
        """quant_grid <- the codebook of the device-side pick (a row of the stacked candidates)."""
        self.quant_grid.data = self._to_grid(np.stack(spec["grids"])).index_select(0, idx)[0]

    def _selectable_rows(self, spec):
        """[ntypes, 2^bit] float32: what quant_grid holds for each candidate type (antq_calibrate_install gathers one row)."""
        g = spec["grids"]
        return np.stack(g) if len({len(v) for v in g}) == 1 else None

    def _after_install_selected(self, spec):
        pass

    def _adopt_plan(self, spec, t):
        self._plan = _lib.plan_for(spec["grids"][t])
        self._gmax = spec["gmaxs"][t]
        self._grid_key = self._grid_now()

    def _calib_apply(self, spec, t, alpha, score, rows):
        """The state _init_quant_para leaves behind, from the batch's results for type t."""
        self.mode = spec["modes"][t]
        self._install_grid(spec["grids"][t])
        self.alpha.data = alpha.clone().unsqueeze(1)
        self._searched = True
        self._mse_later(score, rows)
        self._emit(self._calib_line())
        self.has_inited_quant_para.data = torch.ones_like(self.has_inited_quant_para)
        self._hm_known('has_inited_quant_para', 1.0)
        self._steady = True

    # ---------------------------------------------------------------- steady state
    # (_rest_buffer / _rest_alpha / _at_rest: _mirror.WeightsAtRestMixin)

    def _forward(self, data):
        """AQ:535-551 as one fused kernel."""
        plan = self._ensure_plan()
        return core.fake_quant(data, self._rest_alpha(), plan, self._gmax, self.is_perchannel,
                               unordered=self._at_rest(data), out=self._rest_buffer(data))

    def tensor_forward(self, tensor, input_tensor=None):
        if self.mode == "base":
            return tensor
        if not self.is_enable:
            return tensor
        if self.is_input:
            if not self.is_enable_activation:
                return tensor
        else:
            if not self.is_enable_weight:
                return tensor

        self._before_calibration(tensor)
        self._grad_call = torch.is_grad_enabled() and (tensor.requires_grad or self.alpha.requires_grad)
        self._defer_allowed = not self._grad_call
        with torch.no_grad():
            try:
                self._init_quant_para(tensor, input_tensor)
            except BaseException:
                core.forget_absmax()
                raise
        if self._spec_out is not None:                  # calibrated a moment ago with the pick still on the device
            out, self._spec_out = self._spec_out, None
            return out

        if self._bank is None and self._auto_bank is not None and self._steady and not self.is_input and not self._grad_call:
            self._auto_bank.poke(self)         # (every weight quantiser calibrated: one launch for all of them from now on)
        if self._bank is not None:
            hit = self._bank.lookup(self, tensor, training=self._grad_call)
            if hit is not None:
                return hit

        if self.mode == 'outlier':
            q_tensor = self.outlier_quant(tensor)
        else:
            q_tensor = self._forward(tensor)

        return q_tensor


class TensorQuantizer(Quantizer):
    def __init__(self, **kwargs):
        super(TensorQuantizer, self).__init__(**kwargs)

    def forward(self, tensor, input_tensor=None):
        return self.tensor_forward(tensor, input_tensor)


def _quantizer_pair(owner, mode, wbit, abit, args, operator):
    """weight quantiser: signed, per output channel; input quantiser: unsigned until a negative value is seen
    (update_signed), per tensor -- the pairing every wrapper of the reference uses (AQ:589-590, :627-628)."""
    assert mode is not None, 'Quantizer is not initilized!'
    owner.quant_weight = TensorQuantizer(mode=mode, bit=wbit, is_signed=True, is_enable=True, args=args,
                                         operator=operator)
    owner.quant_input = TensorQuantizer(mode=mode, bit=abit, is_signed=False, is_enable=True, args=args,
                                        operator=operator, is_input=True)


def _adopt_parameters(owner, src, out_channels):
    """Clone weight / bias of the wrapped layer and pre-size the per-channel alpha (AQ:596-607, :634-640)."""
    owner.quant_weight.alpha.data = torch.ones([out_channels, 1])
    owner.weight = nn.Parameter(src.weight.data.clone())
    owner.bias = nn.Parameter(src.bias.data.clone()) if getattr(src, "bias", None) is not None else None


class Conv2dQuantizer(nn.Module):
    """nn.Conv2d with fake-quantised weight and input (AQ:582-617)."""

    def __init__(self, mode=None, wbit=None, abit=None, args=None):
        super(Conv2dQuantizer, self).__init__()
        _quantizer_pair(self, mode, wbit, abit, args, self._conv_forward)

    def set_param(self, conv):
        for name in ("in_channels", "out_channels", "kernel_size", "stride", "padding", "dilation", "groups"):
            setattr(self, name, getattr(conv, name))
        _adopt_parameters(self, conv, conv.out_channels)

    def _conv_forward(self, input, weight):
        return F.conv2d(input, weight, self.bias, self.stride, self.padding, self.dilation, self.groups)

    def forward(self, input):
        if not self.quant_input._steady:
            self.quant_input.prefetch_sign(input)          # (first call: its sign is on its way while the weight calibrates)
        weight = self.quant_weight(self.weight, input)     # re-quantised on every forward, like the reference
        input = self.quant_input(input, self.weight)
        return self._conv_forward(input, weight)


class LinearQuantizer(nn.Module):
    """nn.Linear with fake-quantised weight and input (AQ:620-646)."""

    def __init__(self, mode=None, wbit=None, abit=None, args=None):
        super(LinearQuantizer, self).__init__()
        _quantizer_pair(self, mode, wbit, abit, args, F.linear)

    def set_param(self, linear):
        self.in_features, self.out_features = linear.in_features, linear.out_features
        _adopt_parameters(self, linear, linear.out_features)

    def forward(self, input):
        if not self.quant_input._steady:
            self.quant_input.prefetch_sign(input)
        weight = self.quant_weight(self.weight, input)
        input = self.quant_input(input, self.weight)
        return F.linear(input, weight, self.bias)


# north_star spelling of the two wrappers
QuantConv2d = Conv2dQuantizer
QuantLinear = LinearQuantizer
