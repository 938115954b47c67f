#!/bin/bash
# A/B build of the pair beam kernel: scripts/build_variant.sh NAME [extra hipcc flags]  ->  scripts/_timing/libporeover_hip_NAME.so
# (the other objects are reused from the regular build; select with POREOVER_HIP_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p scripts/_timing
python3 poreover_amd/build.py >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -c poreover_amd/csrc/po_beam2d.hip -o /tmp/po_beam2d_$name.o
objs=$(ls poreover_amd/csrc/_obj/*.o | grep -v "/po_beam2d\.hip\.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/po_beam2d_$name.o -o scripts/_timing/libporeover_hip_$name.so
echo scripts/_timing/libporeover_hip_$name.so
