#!/bin/bash
# Register / scratch / LDS / occupancy of every physics kernel of the HIP library, from the compiler remarks the build keeps
# (build/<variant>/kernel<n>.log, written by __graft_entry__.build_libs; variant = default | d12 | prof | ...).
cd "$(dirname "$0")/.."
v=${1:-default}
cat build/$v/kernel*.log | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/.*remark: [^ ]* //' | paste - - - - - - - | grep jaco_physics
