#!/usr/bin/env python3
"""Small-batch latency of a quantised HF BERT-base forward (batch 1 x 128 tokens, fp32, ANT flint-4 W+A): eager with
quantisation off, eager with quantisation on (146 quantiser launches + WeightBank), and the same forward captured into
a hipGraph -- possible because the calibrated path never reads back from the device."""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from transformers import BertConfig, BertModel  # noqa: E402

from ant_quantization_amd.ant import quant_model as qm, quant_utils as qu  # noqa: E402
from ant_quantization_amd.weight_bank import WeightBank  # noqa: E402

dev = torch.device("cuda:0")


def wall(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    qu.set_quantizer(types.SimpleNamespace(mode="flint", wbit=4, abit=4, w_up=150, a_up=150, w_low=75, a_low=75,
                                           percent=100, search=False))
    torch.manual_seed(0)
    model = qm.quantize_model(BertModel(BertConfig(), add_pooling_layer=False).eval()).to(dev).eval()
    emb = torch.randn(1, 128, 768, device=dev)                   # inputs_embeds: keeps the capture free of id lookups
    mask = torch.ones(1, 128, device=dev)
    fwd = lambda: model(inputs_embeds=emb, attention_mask=mask).last_hidden_state  # noqa: E731
    with torch.no_grad():
        qu.disable_quantization(model)
        t_off = wall(fwd)
        qu.enable_quantization(model)
        fwd()                                                    # calibration
        t_layer = wall(fwd)
        WeightBank(model)
        t_bank = wall(fwd)
        y_eager = fwd().clone()
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fwd()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                y_static = fwd()
            t_graph = wall(graph.replay)
            same = bool(torch.equal(y_static, y_eager))
        except Exception as e:                                   # the HF model may sync on its own
            t_graph, same = float("nan"), "capture failed: %s" % str(e).splitlines()[0][:80]
    print("BERT-base fp32, batch 1 x 128 tokens, ANT flint-4 W+A (146 quantisers)")
    print("  eager, quantisation off                       %7.3f ms" % t_off)
    print("  eager, quantisation on, per-layer weights     %7.3f ms" % t_layer)
    print("  eager, quantisation on, WeightBank            %7.3f ms" % t_bank)
    print("  hipGraph replay, quantisation on, WeightBank  %7.3f ms   (same bits as eager: %s)" % (t_graph, same))


if __name__ == "__main__":
    main()
