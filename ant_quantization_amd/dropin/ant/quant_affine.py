"""Stands in for the reference's antquant/quant_affine.py."""
import _path  # noqa: F401
from ant_quantization_amd.ant.quant_affine import *  # noqa: F401,F403
