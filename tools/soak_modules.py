#!/usr/bin/env python3
"""Soak run of the module path (GPU): N epochs of [train steps with a .data-writing optimiser -> no_grad evaluation] on a
small quantised stack, both trees, fp32 and bf16, fresh models created and dropped along the way.  Watches what must stay
flat after the first epochs: device memory allocated / reserved, the pinned-slot pool, the workspace / plan caches, the
host's resident set.
    python tools/soak_modules.py [epochs]"""
import gc
import importlib
import os
import resource
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
from ant_quantization_amd import _lib, _mirror  # noqa: E402

dev = torch.device("cuda:0")
EPOCHS = int(sys.argv[1]) if len(sys.argv) > 1 else 60


def args_for(mode):
    return types.SimpleNamespace(mode=mode, wbit=4, abit=4, w_up=150, a_up=150, w_low=75, a_low=75, percent=100, search=False, no_outlier=False)


def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


def state():
    return (torch.cuda.memory_allocated(dev) >> 10, torch.cuda.memory_reserved(dev) >> 20, len(_mirror._slots.chunks),
            len(_mirror._slots.free), len(_lib._workspaces))


def one_model(tree, dt, mode, epochs):
    qmod = importlib.import_module("ant_quantization_amd.%s.quant_model" % tree)
    qutil = importlib.import_module("ant_quantization_amd.%s.quant_utils" % tree)
    qutil.set_quantizer(args_for(mode))
    torch.manual_seed(1)
    net = nn.Sequential(*[m for _ in range(6) for m in (nn.Linear(512, 512), nn.GELU())], nn.Linear(512, 8))
    model = qmod.quantize_model(net).to(dev).to(dt)
    qutil.enable_quantization(model)
    x = torch.randn(256, 512, device=dev).to(dt)
    model.eval()
    with torch.no_grad():
        model(x)
    marks = []
    for ep in range(epochs):
        model.train()
        for _ in range(4):
            loss = model(x).float().pow(2).mean()
            loss.backward()
            for p in model.parameters():
                if p.grad is not None:
                    p.data.add_(-1e-3 * p.grad.data)
                    p.grad = None
        model.eval()
        with torch.no_grad():
            for _ in range(3):
                y = model(x)
        assert torch.isfinite(y.float()).all()
        if ep in (epochs // 3, epochs - 1):
            torch.cuda.synchronize()
            marks.append(state())
    del model, net, x, y, loss
    gc.collect()
    return marks


if __name__ == "__main__":
    import contextlib
    import io
    rows = []
    for rnd in range(3):                                  # models come and go: caches keyed on them must not grow
        for tree, mode in (("ant", "ant-int-pot-flint"), ("olive", "ant-int-flint")):
            for dt in (torch.float32, torch.bfloat16):
                with contextlib.redirect_stdout(io.StringIO()):
                    m = one_model(tree, dt, mode, EPOCHS)
                torch.cuda.synchronize()
                rows.append((rnd, tree, str(dt)[6:], m, state(), rss_mb()))
                print("round %d %-5s %-8s  at 1/3: alloc %d KiB reserved %d MiB slots %d chunks / %d free, workspaces %d | at end: alloc %d KiB reserved %d MiB slots %d / %d ws %d | after del: alloc %d KiB | host max RSS %.0f MiB"
                      % (rnd, tree, str(dt)[6:], *m[0], *m[1], rows[-1][4][0], rows[-1][5]), flush=True)
    ok = True
    for r in rows:
        a, b = r[3]
        if b[0] > a[0] + 64 or b[2] != a[2]:              # allocated memory and pinned chunks flat between 1/3 and the end
            ok = False
            print("GROWTH inside a run:", r)
    first = {}
    for r in rows:                                        # and between rounds
        k = (r[1], r[2])
        if k in first:
            if r[4][0] > first[k][4][0] + 64 or r[4][2] > first[k][4][2] or r[5] > first[k][5] + 64:
                ok = False
                print("GROWTH between rounds:", first[k], r)
        else:
            first[k] = r
    print("soak: %s" % ("flat" if ok else "GROWTH"))
    sys.exit(0 if ok else 1)
