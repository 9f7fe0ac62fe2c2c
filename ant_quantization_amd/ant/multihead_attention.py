"""MultiheadAttentionQuantizer (ant_quantization/antquant/multihead_attention.py:486-686).

The reference vendors ~450 lines of torch 1.11's `multi_head_attention_forward` only to place four
quantisers in it: packed in-projection weight and its input, out-projection weight and its input
(:525-529, :459, :663-668), for torchvision's ViT.  The wrapper here keeps the reference's parameter
and quantiser names (state-dict compatibility) and writes the attention itself out in a dozen lines of
plain PyTorch for the case the reference supports: self-attention (`key = value = query`, :665-666)
with equal embedding dims, including the two options of torch's attention its forward passes on (round 5): `add_bias_kv`
(a learnt key / value row appended to the projected sequence, :375-388) and `add_zero_attn` (a zero key / value row
appended per head, :417-425), with or without masks (the reference itself only runs them without: its vendored forward
calls an undefined `pad` as soon as a mask meets either option, :382).  Distinct `kdim` / `vdim` raise here as they do
there (NameError in its set_param, :566-569).  The quantisers are the fused gfx950 ones.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .quant_modules import TensorQuantizer


class MultiheadAttentionQuantizer(nn.Module):
    __constants__ = ['batch_first']

    def __init__(self, mode=None, wbit=None, abit=None, args=None):
        super(MultiheadAttentionQuantizer, self).__init__()
        assert mode is not None, 'Quantizer is not initilized!'
        self.in_quant_weight = TensorQuantizer(mode=mode, bit=wbit, is_signed=True, is_enable=True, args=args)
        self.in_quant_input = TensorQuantizer(mode=mode, bit=abit, is_signed=True, is_enable=True, args=args, is_input=True)
        self.out_quant_weight = TensorQuantizer(mode=mode, bit=wbit, is_signed=True, is_enable=True, args=args)
        self.out_quant_input = TensorQuantizer(mode=mode, bit=abit, is_signed=True, is_enable=True, args=args, is_input=True)

    def set_param(self, MA):
        if not MA._qkv_same_embed_dim:
            raise NotImplementedError("separate q/k/v projection weights (kdim/vdim != embed_dim) are not quantised")
        self.embed_dim, self.num_heads, self.dropout = MA.embed_dim, MA.num_heads, MA.dropout
        self.kdim = self.vdim = MA.embed_dim
        self._qkv_same_embed_dim = True
        self.batch_first = MA.batch_first
        self.head_dim = self.embed_dim // self.num_heads
        self.in_quant_weight.alpha.data = torch.ones([3 * self.embed_dim, 1])
        self.out_quant_weight.alpha.data = torch.ones([self.embed_dim, 1])
        self.in_proj_weight = nn.Parameter(MA.in_proj_weight.data.clone())
        if MA.in_proj_bias is None:
            self.register_parameter('in_proj_bias', None)
        else:
            self.in_proj_bias = nn.Parameter(MA.in_proj_bias.data.clone())
        self.out_proj_weight = nn.Parameter(MA.out_proj.weight.data.clone())
        if MA.out_proj.bias is None:
            self.register_parameter('out_proj_bias', None)
        else:
            self.out_proj_bias = nn.Parameter(MA.out_proj.bias.data.clone())
        if MA.bias_k is not None:                          # add_bias_kv (:598-603)
            self.bias_k = nn.Parameter(MA.bias_k.data.clone())
            self.bias_v = nn.Parameter(MA.bias_v.data.clone())
        else:
            self.bias_k = self.bias_v = None
        self.add_zero_attn = bool(MA.add_zero_attn)

    def forward(self, query, key=None, value=None, key_padding_mask=None, need_weights=True, attn_mask=None,
                average_attn_weights=True):
        w_in = self.in_quant_weight(self.in_proj_weight)
        x = self.in_quant_input(query)                      # key = value = query in the reference (:665-666)
        w_out = self.out_quant_weight(self.out_proj_weight)

        batched = x.dim() == 3
        if not batched:
            x = x.unsqueeze(1)
        elif self.batch_first:
            x = x.transpose(1, 0)
        L, N, E = x.shape
        H, hd = self.num_heads, self.head_dim
        q, k, v = F.linear(x, w_in, self.in_proj_bias).chunk(3, dim=-1)
        if attn_mask is not None and attn_mask.dim() == 2:
            attn_mask = attn_mask.unsqueeze(0)
        if key_padding_mask is not None and not batched:
            key_padding_mask = key_padding_mask.unsqueeze(0)
        if self.bias_k is not None:                        # :375-388: one learnt row behind the keys / values, masks padded
            k = torch.cat([k, self.bias_k.to(k.dtype).repeat(1, N, 1)])
            v = torch.cat([v, self.bias_v.to(v.dtype).repeat(1, N, 1)])
            if attn_mask is not None:
                attn_mask = F.pad(attn_mask, (0, 1))
            if key_padding_mask is not None:
                key_padding_mask = F.pad(key_padding_mask, (0, 1))
        q = q.contiguous().view(L, N * H, hd).transpose(0, 1)
        k = k.contiguous().view(k.shape[0], N * H, hd).transpose(0, 1)
        v = v.contiguous().view(v.shape[0], N * H, hd).transpose(0, 1)
        if self.add_zero_attn:                             # :417-425: a zero row per head
            k = torch.cat([k, torch.zeros((N * H, 1, hd), dtype=k.dtype, device=k.device)], dim=1)
            v = torch.cat([v, torch.zeros((N * H, 1, hd), dtype=v.dtype, device=v.device)], dim=1)
            if attn_mask is not None:
                attn_mask = F.pad(attn_mask, (0, 1))
            if key_padding_mask is not None:
                key_padding_mask = F.pad(key_padding_mask, (0, 1))
        S = k.shape[1]
        scores = torch.bmm(q, k.transpose(1, 2)) / math.sqrt(hd)
        if attn_mask is not None:
            m = attn_mask
            scores = scores.masked_fill(m, float("-inf")) if m.dtype == torch.bool else scores + m
        if key_padding_mask is not None:
            kp = key_padding_mask.view(N, 1, 1, S).expand(-1, H, -1, -1).reshape(N * H, 1, S)
            scores = scores.masked_fill(kp, float("-inf")) if kp.dtype == torch.bool else scores + kp
        attn = F.softmax(scores, dim=-1)
        if self.dropout > 0.0 and self.training:
            attn = F.dropout(attn, p=self.dropout)
        out = torch.bmm(attn, v).transpose(0, 1).contiguous().view(L * N, E)
        out = self.out_quant_input(out)                     # :459
        out = F.linear(out, w_out, self.out_proj_bias).view(L, N, E)
        weights = None
        if need_weights:
            weights = attn.view(N, H, L, S)
            if average_attn_weights:
                weights = weights.mean(dim=1)
            if not batched:
                weights = weights.squeeze(0)
        if not batched:
            out = out.squeeze(1)
        elif self.batch_first:
            out = out.transpose(1, 0)
        return out, weights
