#!/bin/bash
# A/B of the step kernel's workgroup shape (experimental builds, never the product): GTE_WAVES
# wavefronts per workgroup x GTE_GATHER_U loads in flight per lane.  Builds libgte_w<W>u<U>.so next
# to libgte.so; tools/waves_ab.py times them on config 3.
set -e
cd "$(dirname "$0")/../gym-trading-env_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
S="gte_kernels.hip gte_hot.hip gte_hot_nt.hip gte_aux.hip gte_rollout.hip gte_comm.hip gte_api.hip"
for wu in "2 4" "2 8" "3 4" "3 6" "4 4"; do
  set -- $wu
  /opt/rocm/bin/hipcc $F -DGTE_WAVES=$1 -DGTE_GATHER_U=$2 -shared $S -ldl -o libgte_w$1u$2.so &
done
wait
ls -la libgte_w*.so
