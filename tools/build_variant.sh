#!/bin/bash
# Build a named variant of the HIP library into build/ (for A/B timing with tools/gpu_ab.sh); extra args go to hipcc.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build
cd mujoco_jaco_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fgpu-flush-denormals-to-zero -shared -fPIC -I include "$@" \
  -o ../../build/libjaco_env_$name.so jaco_env.hip model_blob.cpp 2>&1 | grep -v "warning\|^$" || true
ls -la ../../build/libjaco_env_$name.so
