#!/bin/bash
# Probe builds of the split weight-gradient launch for tools/run_wgrad_split_abl.sh: booster_gym_amd/libbg_<name>.so = the product objects with
# bg_wgrad_split.hip compiled under -DWS_ABL_* / -DBG_PROBE_NO_STAGE (run `make -C booster_gym_amd/csrc` first; the .so files are git-ignored).
set -e
cd "$(dirname "$0")/../booster_gym_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -fno-signed-zeros -ffinite-math-only -fassociative-math -freciprocal-math -fno-trapping-math"
OBJS="bg_sim.o bg_ppo.o bg_mlp.o bg_mlp_chain.o bg_mlp_chain_split.o bg_mlp_chain_split_bwd.o bg_mlp_split.o bg_head.o bg_wgrad.o bg_tail.o bg_urdf.o bg_model.o"
for v in "base:" "nosplit:-DWS_ABL_NOSPLIT" "nostage:-DBG_PROBE_NO_STAGE" "nosplit_nostage:-DWS_ABL_NOSPLIT -DBG_PROBE_NO_STAGE" "noldsread:-DWS_ABL_NOLDSREAD" \
         "nothing:-DWS_ABL_NOSPLIT -DBG_PROBE_NO_STAGE -DWS_ABL_NOBARRIER -DWS_ABL_NOLDSREAD" "nomfma:-DWS_ABL_NOMFMA" "copyonly:-DWS_ABL_NOMFMA -DWS_ABL_NOSPLIT -DWS_ABL_NOLDSREAD"; do
    n=${v%%:*}; fl=${v#*:}
    /opt/rocm/bin/hipcc $F $fl -c bg_wgrad_split.hip -o /tmp/ws_$n.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbg_$n.so $OBJS /tmp/ws_$n.o
done
ls ../libbg_*.so
