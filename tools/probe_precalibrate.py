"""First (calibrating) forward of a quantised BERT-base with the weights searched in one batch before the first layer
(weight_bank.AutoBank.precalibrate, antq_calibrate_batch) against the per-layer schedule: wall time, host waits, memo hits.
    python tools/probe_precalibrate.py [mode=ant-int-pot-flint]"""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transformers import BertConfig, BertModel
from ant_quantization_amd.ant import quant_model as qm, quant_utils as qu
from ant_quantization_amd import weight_bank
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "ant-int-pot-flint"
args = types.SimpleNamespace(mode=mode, wbit=4, abit=4, w_up=150, a_up=150, w_low=75, a_low=75, percent=100, search=False)
qu.set_quantizer(args)
torch.manual_seed(0)
ids = torch.randint(0, 30000, (64, 128), device=dev)
orig = weight_bank.AutoBank.precalibrate
def timed(self):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    orig(self)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    if self.precalibrated: print("  precalibrate: host %.1f ms, +drain %.1f ms (%d quantisers)" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, self.precalibrated), file=sys.__stdout__)
weight_bank.AutoBank.precalibrate = timed
waits = {"n": 0, "t": 0.0, "ev_n": 0, "ev_t": 0.0}
_cpu, _evs = torch.Tensor.cpu, torch.cuda.Event.synchronize
def cpu(self, *a, **k):
    t0 = time.perf_counter(); r = _cpu(self, *a, **k); waits["n"] += 1; waits["t"] += time.perf_counter() - t0; return r
def evs(self):
    t0 = time.perf_counter(); r = _evs(self); waits["ev_n"] += 1; waits["ev_t"] += time.perf_counter() - t0; return r
torch.Tensor.cpu, torch.cuda.Event.synchronize = cpu, evs
for setting in (1, 0, 1, 0):
    model = qm.quantize_model(BertModel(BertConfig()).eval()).to(dev).eval()
    with torch.no_grad():
        qu.disable_quantization(model)
        model(ids); model(ids)
        torch.cuda.synchronize()
        qu.enable_quantization(model)
        model._antq_auto_bank.batch_calibration = setting
        sys.stdout = open(os.devnull, "w")
        for k in waits: waits[k] = 0
        t0 = time.perf_counter()
        model(ids)
        torch.cuda.synchronize()
        sys.stdout = sys.__stdout__
        print("mode %s batch_calibration=%d: first forward %.1f ms; .cpu() %d waits %.1f ms, event %d waits %.1f ms, memo hits %d" % (mode, setting, (time.perf_counter() - t0) * 1e3, waits["n"], waits["t"] * 1e3, waits["ev_n"], waits["ev_t"] * 1e3, __import__("ant_quantization_amd").core.search_memo.hits))
