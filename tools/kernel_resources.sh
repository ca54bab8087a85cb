#!/bin/bash
# Register / scratch / LDS / occupancy of every kernel of the HIP library (compiler remarks of a device-only compile).
cd "$(dirname "$0")/../mujoco_jaco_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fgpu-flush-denormals-to-zero -I include "$@" \
  --cuda-device-only -c -Rpass-analysis=kernel-resource-usage jaco_env.hip -o /tmp/jaco_env_dev.o 2>&1 | \
  grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/.*remark: [^ ]* //' | paste - - - - - - - | grep jaco_physics
