#!/bin/bash
# builds tools/ubench (dev micro-benchmark) against the in-tree libantq.so
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench.hip -o tools/ubench \
  -Lant_quantization_amd -lantq -Wl,-rpath,'$ORIGIN/../ant_quantization_amd'
