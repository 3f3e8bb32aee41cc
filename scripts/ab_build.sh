#!/bin/bash
# A/B builds of libvine_hip.so with extra compile flags: scripts/ab_build.sh <name> "<flags>"  -> build/libvine_<name>.so
# (run with VINE_HIP_LIB=build/libvine_<name>.so; experiments only)
set -e
mkdir -p build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=fast -fno-slp-vectorize -Wall -Wno-unused-function $2 \
    -o build/libvine_$1.so vine_robot_isaacgymenvs_amd/csrc/vine_hip.hip vine_robot_isaacgymenvs_amd/csrc/ppo_kernels.hip
echo built build/libvine_$1.so
