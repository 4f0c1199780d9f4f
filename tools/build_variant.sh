#!/bin/bash
# builds variants/libgpbc_<name>.so for kernel-tuning A/B runs (tools/variant_bench.sh).  The shipped sources carry ONE code path, so a
# variant is a COPY of csrc/ with the experiment applied:   cp -r gopairingbasedcryptography_amd/csrc variants/csrc_x; edit; then
# usage: GPBC_SRC=variants/csrc_x bash tools/build_variant.sh <name> [extra compiler flags...]      (GPBC_SRC defaults to the tree's csrc/)
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$(cd "${GPBC_SRC:-$R/gopairingbasedcryptography_amd/csrc}" && pwd)
T=$(mktemp -d)
for s in gpbc_core gpbc_pairing gpbc_curve gpbc_wire gpbc_msm; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -I$R/include "$@" -c $SRC/$s.hip -o $T/$s.o &
done
wait
mkdir -p $R/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/variants/libgpbc_$NAME.so $T/*.o
rm -rf $T
echo $R/variants/libgpbc_$NAME.so
