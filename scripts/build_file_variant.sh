#!/bin/bash
# A/B build of ONE source file: scripts/build_file_variant.sh FILE NAME [extra hipcc flags]  ->  scripts/variants/libporeover_hip_NAME.so
# (FILE without directory or suffix, e.g. po_pair; the other objects are reused from the regular build; select with POREOVER_HIP_LIB=...)
set -e
cd "$(dirname "$0")/.."
file=$1; name=$2; shift 2
mkdir -p scripts/variants
python3 poreover_amd/build.py >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -c poreover_amd/csrc/$file.hip -o /tmp/${file}_$name.o
# (the variant object of po_beam2d_reg is the whole file — PO_REG_TU undefined —, so the regular build's second object of it stays out too)
objs=$(ls poreover_amd/csrc/_obj/*.o | grep -v "/$file\.hip\.o" | grep -v "/${file}_wide\.hip\.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/${file}_$name.o -o scripts/variants/libporeover_hip_$name.so
echo scripts/variants/libporeover_hip_$name.so
