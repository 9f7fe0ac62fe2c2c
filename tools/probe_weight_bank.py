#!/usr/bin/env python3
"""Module-level cost of the weight schedules on a random-init BERT-base (HF, batch 64 x 128 tokens) and a
ResNet-50 (torchvision's architecture written out here: no torchvision in the image; batch 64 x 3 x 224 x 224), ANT flint-4
W+A: forward time with quantisation off / per-layer weight launches (the reference's schedule, set_weight_bank(model, False))
/ the DEFAULT (one batched refresh per no-grad forward) / resident (set_weights_at_rest: zero weight launches at rest) /
the default captured into a hipGraph.  Done = the default within 1 % of the resident mode."""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from ant_quantization_amd.ant import quant_model as qm, quant_utils as qu  # noqa: E402

dev = torch.device("cuda:0")


class Bottleneck(nn.Module):
    def __init__(self, inp, planes, stride, down):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv2d(inp, planes, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
        self.down = nn.Sequential(nn.Conv2d(inp, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4)) if down else None

    def forward(self, x):
        y = torch.relu(self.bn1(self.conv1(x)))
        y = torch.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return torch.relu(y + (x if self.down is None else self.down(x)))


class ResNet50(nn.Module):
    def __init__(self):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1))
        blocks, inp = [], 64
        for planes, n, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
            for b in range(n):
                blocks.append(Bottleneck(inp, planes, stride if b == 0 else 1, b == 0))
                inp = planes * 4
        self.blocks = nn.Sequential(*blocks)
        self.fc = nn.Linear(2048, 1000)

    def forward(self, x):
        return self.fc(torch.flatten(nn.functional.adaptive_avg_pool2d(self.blocks(self.stem(x)), 1), 1))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


def run(name, base, inp, dtype):
    args = types.SimpleNamespace(mode="flint", wbit=4, abit=4, w_up=150, a_up=150, w_low=75, a_low=75, percent=100, search=False)
    qu.set_quantizer(args)
    model = qm.quantize_model(base).to(dev).to(dtype).eval()
    if inp.is_floating_point():
        inp = inp.to(dtype)
    with torch.no_grad():
        qu.disable_quantization(model)
        t_off = timed(lambda: model(inp))
        qu.enable_quantization(model)
        model(inp)                                   # calibration
        model(inp)                                   # the bank attaches
        ab = model._antq_auto_bank
        bank = ab.bank
        assert bank is not None and not bank.resident
        # default and resident alternate (5 rounds, best of each): a host-bound forward drifts by several % over seconds
        t_default = t_resident = 1e9
        per_fwd = 0.0
        for _ in range(5):
            qu.set_weights_at_rest(model, False)
            model(inp)
            n0 = bank.launches
            t_default = min(t_default, timed(lambda: model(inp)))
            per_fwd = (bank.launches - n0) / (3 + 3 * 20)
            qu.set_weights_at_rest(model, True)
            model(inp)
            n1 = bank.launches
            t_resident = min(t_resident, timed(lambda: model(inp)))
            assert bank.launches == n1
        qu.set_weights_at_rest(model, False)
        qu.set_weight_bank(model, False)
        t_layer = timed(lambda: model(inp))
        qu.set_weight_bank(model, True)
        model(inp)
        bank = ab.bank
        # the default schedule inside a hipGraph
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            model(inp)
        torch.cuda.current_stream().wait_stream(side)
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                model(inp)
            t_graph = timed(g.replay)
        except Exception as ex:        # noqa: BLE001  (a model whose own forward cannot be captured)
            print("  (graph capture failed: %s)" % (str(ex).splitlines() or ["?"])[0])
            t_graph = float("nan")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            bank.refresh()
        torch.cuda.synchronize()
        t_refresh = (time.perf_counter() - t0) / 50 * 1e3
    nw = sum(e["out"].numel() for e in bank.entries.values())
    print("%s %s, ANT flint-4 W+A, %d weight quantisers (%.1f M weights), %d skipped" % (
        name, str(dtype)[6:], len(bank.entries), nw / 1e6, len(bank.skipped)))
    print("  forward, quantisation off                                   %8.3f ms" % t_off)
    print("  forward, per-layer weight launches (set_weight_bank False)   %8.3f ms" % t_layer)
    print("  forward, DEFAULT: one batched refresh per forward            %8.3f ms   (%.2f bank launches per forward)" % (t_default, per_fwd))
    print("  forward, resident (set_weights_at_rest: 0 weight launches)   %8.3f ms   -> default / resident = %.4f" % (
        t_resident, t_default / t_resident))
    print("  forward, DEFAULT captured into a hipGraph (replay)           %8.3f ms" % t_graph)
    print("  one bank refresh alone                                       %8.3f ms = %.0f Gelem/s" % (t_refresh, nw / t_refresh / 1e6), flush=True)


def main():
    from transformers import BertConfig, BertModel
    torch.manual_seed(0)
    ids = torch.randint(0, 30000, (64, 128), device=dev)
    for dt in (torch.float32, torch.bfloat16):
        run("BERT-base, batch 64 x 128 tokens,", BertModel(BertConfig()).eval(), ids, dt)
    x = torch.randn(64, 3, 224, 224, device=dev)
    for dt in (torch.float32, torch.bfloat16):
        run("ResNet-50, batch 64 x 3 x 224 x 224,", ResNet50().eval(), x, dt)


if __name__ == "__main__":
    main()
