#!/bin/bash
# usage: tools/build_variant.sh <name> <extra hipcc flags...>  ->  rl_ptg_amd/lib/exp/libptg_env_<name>.so  (load it with PTG_LIB_PATH=...)
# an experiment build of the one source with extra -D flags (ablations, stamps); never the product library
name=$1; shift
D=rl_ptg_amd/lib/exp; mkdir -p $D/obj_$name
pids=()
for k in 0 1 2 3 4 5 6 7 8; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DPTG_PART=$k "$@" -c -o $D/obj_$name/p$k.o rl_ptg_amd/csrc/ptg_env.hip &
  pids+=($!)
done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc -eq 0 ] && /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $D/libptg_env_$name.so $D/obj_$name/p*.o && echo built $D/libptg_env_$name.so
