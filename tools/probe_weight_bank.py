#!/usr/bin/env python3
"""Module-level effect of WeightBank on a random-init BERT-base (HF, fp32, batch 64 x 128 tokens, ANT flint-4):
forward time with quantisation off / per-layer weight quantisation (the reference's schedule) / the bank."""
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


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    args = types.SimpleNamespace(mode="flint", wbit=4, abit=4, w_up=150, a_up=150, w_low=75, a_low=75, percent=100,
                                 search=False)
    qu.set_quantizer(args)
    torch.manual_seed(0)
    base = BertModel(BertConfig()).eval()
    model = qm.quantize_model(base).to(dev).eval()
    ids = torch.randint(0, 30000, (64, 128), device=dev)
    with torch.no_grad():
        qu.disable_quantization(model)
        t_off = timed(lambda: model(ids))
        qu.enable_quantization(model)
        t0 = time.perf_counter()
        model(ids)
        torch.cuda.synchronize()
        t_cal = (time.perf_counter() - t0) * 1e3
        t_layer = timed(lambda: model(ids))
        bank = WeightBank(model)
        t_bank = timed(lambda: model(ids))
        n0 = bank.launches

        def step():
            bank.invalidate()
            model(ids)
        t_bank_refresh = timed(step)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            bank.refresh()
        torch.cuda.synchronize()
        t_refresh = (time.perf_counter() - t0) / 20 * 1e3
    nw = sum(e["out"].numel() for e in bank.entries.values())
    print("BERT-base fp32, batch 64x128, ANT flint-4 W+A, %d weight quantisers (%.1f M weights), %d skipped" % (
        len(bank.entries), nw / 1e6, len(bank.skipped)))
    print("  forward, quantisation off                       %8.2f ms" % t_off)
    print("  first forward (calibration of all quantisers)   %8.2f ms" % t_cal)
    print("  forward, per-layer weight quantisation           %8.2f ms" % t_layer)
    print("  forward, WeightBank resident (0 weight launches) %8.2f ms   (bank launches so far: %d)" % (t_bank, n0))
    print("  forward, WeightBank refreshed every step         %8.2f ms" % t_bank_refresh)
    print("  one bank refresh alone                           %8.3f ms = %.0f Gelem/s" % (t_refresh, nw / t_refresh / 1e6))


if __name__ == "__main__":
    main()
