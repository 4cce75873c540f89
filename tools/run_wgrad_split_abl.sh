# ablation builds of the split weight-gradient launch, alone (timing only: the results of the ablated builds are wrong by construction)
for n in base nosplit nostage nosplit_nostage noldsread nothing base; do echo -n "$n  "; BG_LIB=$PWD/booster_gym_amd/libbg_$n.so timeout -k 10 200 python tools/wgrad_split_error_probe.py 2>&1 | grep "split9_us" | cut -c1-60; done
