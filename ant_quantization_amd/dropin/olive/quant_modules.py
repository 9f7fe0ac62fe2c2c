"""Stands in for the reference's antquant/quant_modules.py: `sys.path.append(<this directory>)` where the harnesses append
"../antquant" (ImageNet/main.py:14-16, BERT/run_glue.py:45-47, llm/run_clm.py:56-59)."""
import _path  # noqa: F401
from ant_quantization_amd.olive.quant_modules import *  # noqa: F401,F403
