"""ResNet-50's 54 fp32 weights in ONE batched launch (33 us): which part of the gap to a flat single launch is the batch's own?
Variants: without conv1 (ragged K = 147: element-granular blocks), big tensors only, jobs sorted by size, a flat buffer of the
same bytes as one launch, and the copy kernel over the same bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from ant_quantization_amd import _lib, grids
from bench_configs import resnet50_shapes, timed
dev = torch.device("cuda:0")
plan = _lib.plan_for(grids.ant_flint(4, True))
torch.manual_seed(1)
ws = [torch.randn(s[0], int(np.prod(s[1:])), device=dev) * 0.05 for s in resnet50_shapes()]
outs = [torch.empty_like(w) for w in ws]
al = [_lib.absmax(w, w.shape[0], w.shape[1]) for w in ws]
def run(tag, idx):
    jobs = [(ws[i], outs[i], al[i], plan, 10.0, ws[i].shape[0], ws[i].shape[1], True) for i in idx]
    n = sum(ws[i].numel() for i in idx)
    b = _lib.Batch(jobs)
    t = min(timed(b.run, 200) for _ in range(3))
    print("%-52s %3d jobs %6.1f MB  %6.2f us  %5.1f %%" % (tag, len(idx), n * 4 / 1e6, t * 1e6, n * 8 / t / 8e12 * 100), flush=True)
allj = list(range(len(ws)))
run("all 54", allj)
run("without conv1 (K = 147)", allj[1:])
run("tensors >= 1 MB only", [i for i in allj if ws[i].numel() * 4 >= 1 << 20])
run("all, largest first", sorted(allj, key=lambda i: -ws[i].numel()))
run("all, smallest first", sorted(allj, key=lambda i: ws[i].numel()))
flat = torch.cat([w.reshape(-1) for w in ws]).contiguous(); fo = torch.empty_like(flat)
n = flat.numel()
rows = n // 2048
af = _lib.absmax(flat[:rows * 2048], rows, 2048)
t = min(timed(lambda: _lib.fakequant(flat[:rows * 2048], af, plan, 10.0, rows, 2048, True, out=fo[:rows * 2048]), 200) for _ in range(3))
print("%-52s %3d jobs %6.1f MB  %6.2f us  %5.1f %%" % ("flat buffer, rows of 2048, ONE per-tensor launch", 1, n * 4 / 1e6, t * 1e6, n * 8 / t / 8e12 * 100))
b = _lib.Batch([(flat[:rows * 2048], fo[:rows * 2048], af, plan, 10.0, rows, 2048, True)])
t = min(timed(b.run, 200) for _ in range(3))
print("%-52s %3d jobs %6.1f MB  %6.2f us  %5.1f %%" % ("flat buffer, rows of 2048, batched launch of 1 job", 1, n * 4 / 1e6, t * 1e6, n * 8 / t / 8e12 * 100))
t = min(timed(lambda: _lib.copy(flat, fo), 200) for _ in range(3))
print("%-52s %3d jobs %6.1f MB  %6.2f us  %5.1f %%" % ("copy kernel, same bytes", 1, n * 4 / 1e6, t * 1e6, n * 8 / t / 8e12 * 100))
# uniform batches of the same total size: is it the job count, or the row lengths that are no power of two (alpha index by f64 quotient)?
for rows_, k_, nj in ((512, 1024, 50), (512, 1152, 44), (256, 2304, 44), (2048, 256, 50), (64, 64, 2000)):
    xs = [torch.randn(rows_, k_, device=dev) * 0.05 for _ in range(nj)]
    os_ = [torch.empty_like(x) for x in xs]
    as_ = [_lib.absmax(x, rows_, k_) for x in xs]
    b = _lib.Batch([(x, o, a, plan, 10.0, rows_, k_, True) for x, o, a in zip(xs, os_, as_)])
    n = nj * rows_ * k_
    t = min(timed(b.run, 200) for _ in range(3))
    print("%-52s %3d jobs %6.1f MB  %6.2f us  %5.1f %%" % ("uniform %d x %d" % (rows_, k_), nj, n * 4 / 1e6, t * 1e6, n * 8 / t / 8e12 * 100), flush=True)
# the same uniform jobs as views of ONE slab (inputs, outputs, or both): placement of the caller's buffers, not the kernel
for rows_, k_, nj in ((512, 1024, 50), (256, 2304, 44)):
    n1 = rows_ * k_
    for slab_in, slab_out in ((True, True), (False, True), (True, False)):
        si, so = torch.randn(nj, rows_, k_, device=dev) * 0.05, torch.empty(nj, rows_, k_, device=dev)
        xs = [si[i] for i in range(nj)] if slab_in else [si[i].clone() for i in range(nj)]
        os_ = [so[i] for i in range(nj)] if slab_out else [torch.empty(rows_, k_, device=dev) for i in range(nj)]
        as_ = [_lib.absmax(x, rows_, k_) for x in xs]
        b = _lib.Batch([(x, o, a, plan, 10.0, rows_, k_, True) for x, o, a in zip(xs, os_, as_)])
        t = min(timed(b.run, 200) for _ in range(3))
        print("%-52s %3d jobs %6.1f MB  %6.2f us  %5.1f %%" % ("uniform %d x %d, slab in %d out %d" % (rows_, k_, slab_in, slab_out), nj, nj * n1 * 4 / 1e6, t * 1e6, nj * n1 * 8 / t / 8e12 * 100), flush=True)
# ResNet-50 itself with inputs and outputs carved from two slabs (256-byte aligned pieces, the model's order)
def carve(total_like):
    offs, o = [], 0
    for w in ws:
        offs.append(o); o += (w.numel() + 63) // 64 * 64
    return offs, o
offs, tot = carve(ws)
sin, sout = torch.empty(tot, device=dev), torch.empty(tot, device=dev)
xs = []
for w, o in zip(ws, offs):
    v = sin[o:o + w.numel()].view(w.shape); v.copy_(w); xs.append(v)
os_ = [sout[o:o + w.numel()].view(w.shape) for w, o in zip(ws, offs)]
b = _lib.Batch([(x, o, a, plan, 10.0, x.shape[0], x.shape[1], True) for x, o, a in zip(xs, os_, al)])
t = min(timed(b.run, 200) for _ in range(3))
n = sum(w.numel() for w in ws)
print("%-52s %3d jobs %6.1f MB  %6.2f us  %5.1f %%" % ("ResNet-50, inputs and outputs carved from two slabs", 54, n * 4 / 1e6, t * 1e6, n * 8 / t / 8e12 * 100))
b = _lib.Batch([(x, o, a, plan, 10.0, x.shape[0], x.shape[1], True) for x, o, a in zip(ws, os_, al)])
t = min(timed(b.run, 200) for _ in range(3))
print("%-52s %3d jobs %6.1f MB  %6.2f us  %5.1f %%" % ("ResNet-50, outputs only from a slab", 54, n * 4 / 1e6, t * 1e6, n * 8 / t / 8e12 * 100))
