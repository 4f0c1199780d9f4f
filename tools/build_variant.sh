#!/bin/bash
# builds variants/libgpbc_<name>.so from the current tree with extra compiler flags (kernel-tuning A/B runs: tools/variant_bench.sh)
# usage: bash tools/build_variant.sh <name> [flags...]      e.g.  bash tools/build_variant.sh stackargs -DGPBC_F2_ARGS_ON_STACK
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
for s in gpbc_core gpbc_pairing gpbc_curve gpbc_wire gpbc_msm; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c $R/gopairingbasedcryptography_amd/csrc/$s.hip -o $T/$s.o &
done
wait
mkdir -p $R/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/libgpbc_$NAME.so $T/*.o
rm -rf $T
echo $R/variants/libgpbc_$NAME.so
