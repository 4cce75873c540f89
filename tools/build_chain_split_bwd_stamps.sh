#!/bin/bash
# Builds tools/probe/libbg_bwd_stamps.so: the product library with bg_mlp_chain_split_bwd.hip compiled -DBG_CHAIN_PROBE_STAMPS (shader-clock stamps of every wave
# around every chunk barrier of the chained split-bf16 backward kernel), for tools/chain_split_bwd_stamps.py.  EXTRA / NAME: ablation builds (timing only).
#   gpurun -- 'BG_LIB=$GRAFT_REPO_ROOT/tools/probe/libbg_bwd_stamps.so python tools/chain_split_bwd_stamps.py'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/booster_gym_amd/csrc && make
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -fassociative-math -freciprocal-math -fno-trapping-math -mllvm -amdgpu-sched-strategy=max-ilp"
NAME=${NAME:-libbg_bwd_stamps}
/opt/rocm/bin/hipcc $FL -DBG_CHAIN_PROBE_STAMPS $EXTRA -c bg_mlp_chain_split_bwd.hip -o /tmp/$NAME.o
OBJS=$(ls *.o | grep -v bg_mlp_chain_split_bwd.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/probe/$NAME.so $OBJS /tmp/$NAME.o
echo built $R/tools/probe/$NAME.so
