"""`import quant_cuda` as the reference does (ant_quantization/antquant/quant_modules.py:7, built there from
ant_quantization/quant/quant.cpp:27-29): put this directory on sys.path instead of installing the CUDA extension."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)
from ant_quantization_amd.quant_cuda import quant  # noqa: E402,F401
