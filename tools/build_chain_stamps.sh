#!/bin/bash
# Builds tools/probe/libbg_chain_stamps.so: the product library with bg_mlp_chain.hip compiled -DBG_CHAIN_PROBE_STAMPS (a shader-clock stamp of every wave behind
# every chunk barrier of the chained forward kernel), for tools/mlp_chain_stamps.py.  Run here (hipcc cross-compiles), then
#   gpurun -- 'BG_LIB=$GRAFT_REPO_ROOT/tools/probe/libbg_chain_stamps.so python tools/mlp_chain_stamps.py'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/booster_gym_amd/csrc && make
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -fassociative-math -freciprocal-math -fno-trapping-math -mllvm -amdgpu-sched-strategy=max-ilp"
/opt/rocm/bin/hipcc $FL -DBG_CHAIN_PROBE_STAMPS -c bg_mlp_chain.hip -o /tmp/bg_chain_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probe/libbg_chain_stamps.so bg_sim.o bg_ppo.o bg_mlp.o /tmp/bg_chain_stamps.o bg_mlp_split.o bg_head.o bg_wgrad.o bg_wgrad_split.o bg_urdf.o
echo built $R/tools/probe/libbg_chain_stamps.so
