#!/bin/bash
# Build a named variant of the HIP library (mujoco_jaco_amd/libjaco_env_<name>.so; JACO_ENV_LIB=libjaco_env_<name>.so selects it for
# A/B timing); extra args go to hipcc.  Objects and compiler remarks under build/<name>/.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
python __graft_entry__.py variant "$name" "$@"
ls -la mujoco_jaco_amd/libjaco_env_$name.so
