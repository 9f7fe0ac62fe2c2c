"""MI355X-native mirror of olive_quantization/antquant/quant_modules.py (OliVe, ISCA'23).

Differences from the ANT quantiser that matter to the kernels:
  * codebooks normalised so that the outlier threshold is 32 (int / flint), plus the
    abfloat `outliers` codebook; the kernel scans cat(quant_grid, outliers)  (OQ:303-306);
  * scale = alpha / max(quant_grid) -- the NORMAL grid only (OQ:296);
  * outlier-victim pairs on the flattened tensor after the nearest-value step (OQ:311-320):
    done inside the same kernel, each lane owns whole (2k, 2k+1) pairs;
  * clip search starts from the 3-sigma rule, candidates i in range(lb, ub, 2) (OQ:189-233);
  * PTQ only: the whole quantiser is `torch.no_grad()`.

The reference runs ~17 PyTorch launches (~100 B/elem) per forward; here it is one kernel
moving 8 B/elem (fp32) or 4 B/elem (bf16).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib, core, grids
from .._mirror import CalibrationMixin, HostMirrorMixin, WeightsAtRestMixin


class QuantBase():
    """OQ:8-21 (reshape(-1) instead of view(-1): non-contiguous inputs are accepted)."""

    def _quantization(x, quant_grid, plan=None):
        """The operator boundary (AQ:12-18 / OQ:9-15): flat view, the nearest-value operator, reshape back -- what
        `quant_cuda.quant` does, minus the all-zero index tensor the reference allocates and throws away.  `plan`: a
        caller's plan of `quant_grid`, passed to the kernel as a hint it verifies against the buffer."""
        flat = x.reshape(-1).contiguous()
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

        self.w_up = self.args.w_up
        self.a_up = self.args.a_up
        self.w_low = self.args.w_low
        self.a_low = self.args.a_low

        # registration order / names of OQ:41-45
        self.alpha = nn.Parameter(torch.tensor(1.0, requires_grad=True))
        self.register_buffer('bit', torch.tensor(bit))
        self.register_buffer('has_inited_quant_para', torch.tensor(0.0))
        self.register_buffer('quant_grid', torch.ones(2 ** bit))
        self.register_buffer('outliers', torch.ones(2 ** bit))
        self._hm_setup(bit=int(bit), has_inited_quant_para=0.0)
        self.percent = self.args.percent / 100
        self.is_perchannel = True
        if is_input:
            self.is_perchannel = False
        self.search = args.search
        self.mse = torch.tensor(0.0)
        self.name = None

        self._steady = False
        self._plan = None
        self._gmax = 32.0
        self._grid_key = None
        self._searched = False
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
        self._steady = False
        self._plan = None

    def rearm(self):
        """Force the next forward to re-read `bit` / `has_inited_quant_para` / `quant_grid` from the device, whatever the
        buffers' (data_ptr, _version) keys say (set_8_bit_layer_*, and any outside edit through `.data`)."""
        self._steady = False
        self._plan = None
        for ent in self._hm.values():
            ent[0] = None

    @property
    def _no_outlier(self):
        return bool(getattr(self.args, "no_outlier", False))

    # ---------------------------------------------------------------- codebooks (OQ:72-179)
    def _bits(self):
        return int(self._hm_get('bit'))

    def _to_grid(self, values):
        return core.device_grid(values, self.quant_grid.device)

    @torch.no_grad()
    def int_value(self):
        return self._to_grid(grids.olive_int(self._bits(), self.is_signed))

    @torch.no_grad()
    def flint_value(self, exp_base=0):
        return self._to_grid(grids.olive_flint(self._bits(), self.is_signed))

    @torch.no_grad()
    def outlier_value(self, exp_bit=2, exp_base=5):
        return self._to_grid(grids.olive_outliers(self._bits(), self.is_signed, exp_bit, exp_base))

    def _install(self, normal, outl):
        """quant_grid / outliers <- host arrays; the kernel's grid is their concatenation."""
        normal = np.ascontiguousarray(normal, dtype=np.float32)
        outl = np.ascontiguousarray(outl, dtype=np.float32)
        self.quant_grid.data = self._to_grid(normal)
        self.outliers.data = self._to_grid(outl)
        self._set_plan(normal, outl)
        self._grid_key = self._grid_now()

    def _grid_now(self):
        g, o = self.quant_grid, self.outliers
        return (g.data_ptr(), g._version, o.data_ptr(), o._version)

    def _set_plan(self, normal, outl):
        full = normal if self._no_outlier else np.concatenate([normal, outl])
        self._plan = _lib.plan_for(full)
        self._gmax = float(np.max(normal))

    def _ensure_plan(self):
        """Plan of cat(quant_grid, outliers) as they are NOW: the buffers are watched by (data_ptr, _version)."""
        if self._plan is None or self._grid_key != self._grid_now():
            self._set_plan(self.quant_grid.detach().float().cpu().numpy(),
                           self.outliers.detach().float().cpu().numpy())
            self._grid_key = self._grid_now()
        return self._plan

    # ---------------------------------------------------------------- calibration
    @torch.no_grad()
    def mse_loss(self, quant_tensor, source_tensor, p=2.0, is_perchannel=True):
        if is_perchannel:
            return (quant_tensor - source_tensor).abs().pow(p).view(quant_tensor.shape[0], -1).mean(-1).unsqueeze(1)
        return (quant_tensor - source_tensor).abs().pow(p).mean()

    def _three_sigma(self, tensor, per_channel):
        """OQ:193-197 / :213-218: x_max = max(|mean + 3 std|, |mean - 3 std|) (unbiased std), or the abs-max when
        outliers are disabled.  The reference runs t.mean() and t.std() -- at least three reads of the tensor before the
        search reads it again; here ONE read leaves (sum x, sum x^2) in double per row / tensor (antq_moments, fixed
        summation order) and antq_xmax_3sigma applies the roundings the reference's dtype would (fp32: mean / std to
        float; fp16 / bf16 as its LLM scripts run: mean, std, 3 * std, sum and difference each rounded to the tensor's
        dtype).  Agrees with torch's own reductions to their summation-order noise (tests: rtol 2e-6 in fp32)."""
        if self._no_outlier:
            return core.row_absmax(tensor, per_channel)
        t = core._calib_view(tensor).detach()
        if not t.is_contiguous():
            t = t.contiguous()
        rows = t.shape[0] if (per_channel and t.dim() > 0) else 1
        if t.dtype not in (torch.float32, torch.bfloat16, torch.float16) or t.numel() == 0:
            raise _lib.AntqError("OliVe calibration: unsupported tensor dtype %s" % t.dtype)
        return _lib.xmax_3sigma(t, rows, t.numel() // rows, per_row=per_channel)

    @torch.no_grad()
    def _clip_search(self, tensor, need_xmax=True):
        """search_mse's device work: (best_score [rows or 1], alpha [rows or 1], x_max), nothing reduced or reshaped yet."""
        per_channel = self.is_perchannel and (not self.is_input)
        x_max = None
        lb = int(self.w_low) if per_channel else int(self.a_low)
        ub = int(self.w_up) if per_channel else int(self.a_up)
        plan = self._ensure_plan()
        hit = self._type_search.get((plan.grid.tobytes(), lb, ub)) if self._type_search else None
        if hit is not None:
            best_score, alpha, ratios = hit       # this very search was part of the type selection's single pass
        else:
            # (a parallel branch's input quantiser may have searched this very tensor a moment ago: core.SearchMemo)
            ovp = not self._no_outlier
            key = ("olive", lb, ub, 2, plan.grid.tobytes(), self._gmax, ovp) if not per_channel else None
            seen = core.search_memo.get(tensor, key) if key else None
            if seen is not None:
                best_score, alpha, ratios = seen[0], seen[1].clone(), seen[2]
            else:
                x_max = self._three_sigma(tensor, per_channel)
                best_score, alpha, ratios = core.clip_search(tensor, x_max, per_channel, lb, ub, 2, plan, self._gmax, ovp=ovp)
                if key:
                    core.search_memo.put(tensor, key, (best_score, alpha.clone(), ratios))
        self._searched = ratios is not None
        if x_max is None and need_xmax:
            x_max = self._three_sigma(tensor, per_channel)
        return best_score, alpha, x_max, per_channel

    @torch.no_grad()
    def search_mse(self, tensor):
        best_score, alpha, x_max, per_channel = self._clip_search(tensor)
        ratio = (alpha / x_max).mean()        # 0-dim tensor: the reference's float, without the sync
        if per_channel:
            return best_score.sum(), alpha.unsqueeze(1), ratio
        return best_score.sum(), alpha.reshape(()), ratio

    @torch.no_grad()
    def search_adaptive_numeric_type(self, data):
        """OQ:235-256: int vs flint, smallest summed best-MSE wins."""
        mode = self.mode
        outl = np.ascontiguousarray(grids.olive_outliers(self._bits(), self.is_signed), dtype=np.float32)
        modes = [t for t in ("int", "flint") if ("-" + t) in mode]
        normals = [np.ascontiguousarray(grids.olive_grid(t, self._bits(), self.is_signed), dtype=np.float32) for t in modes]
        fulls = [n if self._no_outlier else np.concatenate([n, outl]) for n in normals]
        # both codebooks' clip searches on ONE read of the tensor (antq_search_sse_multi); the search on the grid that is
        # installed afterwards is one of them and is not repeated (search_mse looks it up)
        per_channel = self.is_perchannel and (not self.is_input)
        lb = int(self.w_low) if per_channel else int(self.a_low)
        ub = int(self.w_up) if per_channel else int(self.a_up)
        gm = [float(np.max(n)) for n in normals]
        key = (("olive-types", lb, ub, 2, tuple(f.tobytes() for f in fulls), tuple(gm), not self._no_outlier)
               if not per_channel and len(fulls) > 1 else None)
        res = core.search_memo.get(data, key) if key else None
        if res is not None:
            res = [(b, a.clone(), r) for b, a, r in res]
        else:
            res = core.clip_search_types(data, self._three_sigma(data, per_channel), per_channel, lb, ub, 2,
                                         [_lib.plan_for(f) for f in fulls], gm, ovp=not self._no_outlier)
            if key and res is not None:
                core.search_memo.put(data, key, [(b, a.clone(), r) for b, a, r in res])
        mse_list = []
        if res is not None:
            self._type_search = {(f.tobytes(), lb, ub): r for f, r in zip(fulls, res)}
            mse_list = [r[0].sum().reshape(()) for r in res]
        else:
            for t, n in zip(modes, normals):
                self.mode = t
                self._install(n, outl)
                best, _, _ = self.search_mse(data)
                mse_list.append(best.reshape(()))
        self.mode = modes[np.argsort(torch.stack(mse_list).cpu().numpy())[0]]   # one read-back for both types

    @torch.no_grad()
    def _init_quant_para(self, data, data_b):
        """OQ:258-292."""
        if self._steady and self._hm_fresh():
            return
        self._hm_get('bit')
        if self._hm_get('has_inited_quant_para') != 0:
            self._ensure_plan()
            self._steady = True
            return
        self.update_signed(data)
        if "ant-" in self.mode and self._bits() <= 6 and self._defer_ok(data):
            spec = self._calib_spec(data)
            if spec is not None and len(spec["grids"]) > 1:
                return self._calibrate_deferred(data, spec)        # (the type pick stays on the device)
        outl = grids.olive_outliers(self._bits(), self.is_signed)
        self.outliers.data = self._to_grid(outl)

        if self.is_perchannel:
            self.alpha.data = core.row_absmax(data, True).unsqueeze(1)
        else:
            self.alpha.data = core.row_absmax(data, False).reshape(())

        if self._bits() > 6:
            self.mode = 'int'
        elif "ant-" in self.mode:
            self.search_adaptive_numeric_type(data)

        if self.mode not in ("flint", "int"):
            raise RuntimeError("Unsupported mode: " + self.mode)
        self._install(grids.olive_grid(self.mode, self._bits(), self.is_signed), outl)

        # (search_mse without its two log values, the summed score and the mean clip ratio, which nothing here reads)
        best_score, alpha, _, per_channel = self._clip_search(data, need_xmax=False)
        self.alpha.data = alpha.unsqueeze(1) if per_channel else alpha.reshape(())

        # OQ:283-284 runs _forward + mse_loss once more for the log value `mse`: it is the winning candidate's score,
        # formed when somebody reads it
        if self._searched:
            self._mse_later(best_score, self.alpha.numel() if self.is_perchannel else 1)
        else:
            self.mse = self.mse_loss(self._forward(data), data, 2, is_perchannel=self.is_perchannel).mean()
        self._emit(self._calib_line())

        self.has_inited_quant_para.data = torch.ones_like(self.has_inited_quant_para)
        self._hm_known('has_inited_quant_para', 1.0)
        self._steady = True
        self._type_search = None
        core.forget_absmax()

    # ---------------------------------------------------------------- calibrated ahead of the forward (weight_bank.precalibrate)
    def _calib_spec(self, tensor):
        """What _init_quant_para would search for this quantiser (OQ:258-292): candidate types in the order of
        search_adaptive_numeric_type, normal codebooks (+ outliers unless no_outlier), the 3-sigma statistic, step 2 -- or
        None when the quantiser keeps the step-by-step path.  Weight quantisers (signed, per channel):
        weight_bank.AutoBank.precalibrate; input quantisers (per tensor, sign already learnt): the type pick on the device,
        _mirror.CalibrationMixin._calibrate_deferred."""
        per_channel = self.is_perchannel and (not self.is_input)
        if self.mode == "base" or (not self.is_input and not (per_channel and self.is_signed)):
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
            modes = [t for t in ("int", "flint") if ("-" + t) in self.mode]
            if not modes:
                return None
        elif self.mode in ("flint", "int"):
            modes = [self.mode]
        else:
            return None                    # (the step-by-step path raises the reference's error)
        lb, ub = (int(self.w_low), int(self.w_up)) if per_channel else (int(self.a_low), int(self.a_up))
        if not range(lb, ub, 2):
            return None
        outl = np.ascontiguousarray(grids.olive_outliers(bit, signed), dtype=np.float32)
        normals = [np.ascontiguousarray(grids.olive_grid(t, bit, signed), dtype=np.float32) for t in modes]
        fulls = [n if self._no_outlier else np.concatenate([n, outl]) for n in normals]
        return dict(modes=modes, grids=fulls, normals=normals, outl=outl, gmaxs=[float(np.max(n)) for n in normals], lb=lb, ub=ub,
                    step=2, stat="absmax" if self._no_outlier else "3sigma", ovp=not self._no_outlier)

    def _calib_line(self):
        return "%s\t%d-bit \t %s," % (self.mode, self._bits(), self.name)

    def _install_selected(self, spec, idx):
        """quant_grid <- the normal codebook of the device-side pick (a row of the stacked candidates); the outliers are the
        same for every candidate."""
        self.quant_grid.data = self._to_grid(np.stack(spec["normals"])).index_select(0, idx)[0]
        self.outliers.data = self._to_grid(spec["outl"])

    def _selectable_rows(self, spec):
        """[ntypes, n_normal] float32: the NORMAL codebook of each candidate type (the outliers are common to all)."""
        g = spec["normals"]
        return np.stack(g) if len({len(v) for v in g}) == 1 else None

    def _after_install_selected(self, spec):
        self.outliers.data = self._to_grid(spec["outl"])

    def _adopt_plan(self, spec, t):
        self._set_plan(spec["normals"][t], spec["outl"])
        self._grid_key = self._grid_now()

    def _calib_apply(self, spec, t, alpha, score, rows):
        """The state _init_quant_para leaves behind, from the batch's results for type t."""
        self.mode = spec["modes"][t]
        self._install(spec["normals"][t], spec["outl"])
        self.alpha.data = alpha.clone().unsqueeze(1)
        self._searched = True
        self._mse_later(score, rows)
        self._emit(self._calib_line())
        self.has_inited_quant_para.data = torch.ones_like(self.has_inited_quant_para)
        self._hm_known('has_inited_quant_para', 1.0)
        self._steady = True

    # ---------------------------------------------------------------- steady state
    # (_rest_buffer / _rest_alpha / _at_rest: _mirror.WeightsAtRestMixin)

    @torch.no_grad()
    def _forward(self, data, display=False):
        """OQ:294-330 as one fused kernel (nearest over normal||outliers + victim masking)."""
        plan = self._ensure_plan()
        return core.fake_quant(data, self._rest_alpha(), plan, self._gmax, self.is_perchannel, ovp=not self._no_outlier,
                               unordered=self._at_rest(data), out=self._rest_buffer(data))

    def tensor_forward(self, tensor, input_tensor=None):
        """OQ:332-359.  The reference runs it under @torch.no_grad(); whether the CALLER is in a training forward (gradients
        enabled, the weight or alpha a leaf that wants one) is looked at before entering: such a call is followed by an
        optimiser step, so nothing resident is served to it or trusted after it (weight_bank, weights-at-rest)."""
        self._grad_call = torch.is_grad_enabled() and (tensor.requires_grad or self.alpha.requires_grad)
        with torch.no_grad():
            return self._tensor_forward(tensor, input_tensor)

    def _tensor_forward(self, tensor, input_tensor=None):
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
        self._defer_allowed = True                      # (tensor_forward runs under no_grad in this tree: OQ:332)
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
        return self._forward(tensor)


class TensorQuantizer(Quantizer):
    def __init__(self, **kwargs):
        super(TensorQuantizer, self).__init__(**kwargs)

    def forward(self, tensor, input_tensor=None):
        return self.tensor_forward(tensor, input_tensor)


def _make_pair(owner, mode, wbit, abit, args, operator):
    owner.quant_weight = TensorQuantizer(mode=mode, bit=wbit, is_signed=True, is_enable=True, args=args,
                                         operator=operator)
    owner.quant_input = TensorQuantizer(mode=mode, bit=abit, is_signed=False, is_enable=True, args=args,
                                        operator=operator, is_input=True)


def _clone_wb(owner, src):
    owner.weight = nn.Parameter(src.weight.data.clone())
    try:
        owner.bias = nn.Parameter(src.bias.data.clone())
    except AttributeError:
        owner.bias = None


class Conv1dQuantizer(nn.Module):
    """HF GPT-2 `Conv1D` (y = x @ W + b, weight [in, out]) with quantised operands (OQ:358-386)."""

    def __init__(self, mode=None, wbit=None, abit=None, args=None):
        super(Conv1dQuantizer, self).__init__()
        assert mode is not None, 'Quantizer is not initilized!'
        _make_pair(self, mode, wbit, abit, args, self._conv_forward)

    def set_param(self, conv):
        self.nf = conv.nf
        _clone_wb(self, conv)

    def _conv_forward(self, x, weight):
        size_out = x.size()[:-1] + (self.nf,)
        x = torch.addmm(self.bias, x.view(-1, x.size(-1)), weight)
        return x.view(size_out)

    def forward(self, input):
        if not self.quant_input._steady:
            self.quant_input.prefetch_sign(input)          # (first call: its sign is on its way while the weight calibrates)
        weight = self.quant_weight(self.weight, input)
        input = self.quant_input(input, self.weight)
        return self._conv_forward(input, weight)


class Conv2dQuantizer(nn.Module):
    """OQ:389-423."""

    def __init__(self, mode=None, wbit=None, abit=None, args=None):
        super(Conv2dQuantizer, self).__init__()
        assert mode is not None, 'Quantizer is not initilized!'
        _make_pair(self, mode, wbit, abit, args, self._conv_forward)

    def set_param(self, conv):
        self.in_channels = conv.in_channels
        self.out_channels = conv.out_channels
        self.quant_weight.alpha.data = torch.ones([self.out_channels, 1])
        self.kernel_size = conv.kernel_size
        self.stride = conv.stride
        self.padding = conv.padding
        self.dilation = conv.dilation
        self.groups = conv.groups
        _clone_wb(self, conv)

    def _conv_forward(self, input, weight):
        return F.conv2d(input, weight, self.bias, self.stride, self.padding, self.dilation, self.groups)

    def forward(self, input):
        if not self.quant_input._steady:
            self.quant_input.prefetch_sign(input)          # (first call: its sign is on its way while the weight calibrates)
        weight = self.quant_weight(self.weight, input)
        input = self.quant_input(input, self.weight)
        return self._conv_forward(input, weight)


class LinearQuantizer(nn.Module):
    """OQ:426-450."""

    def __init__(self, mode=None, wbit=None, abit=None, args=None):
        super(LinearQuantizer, self).__init__()
        assert mode is not None, 'Quantizer is not initilized!'
        _make_pair(self, mode, wbit, abit, args, F.linear)

    def set_param(self, linear):
        self.in_features = linear.in_features
        self.out_features = linear.out_features
        self.quant_weight.alpha.data = torch.ones([self.out_features, 1])
        _clone_wb(self, linear)

    def forward(self, input):
        if not self.quant_input._steady:
            self.quant_input.prefetch_sign(input)          # (first call: its sign is on its way while the weight calibrates)
        weight = self.quant_weight(self.weight, input)
        input = self.quant_input(input, self.weight)
        return F.linear(input, weight, self.bias)


QuantConv2d = Conv2dQuantizer
QuantLinear = LinearQuantizer
