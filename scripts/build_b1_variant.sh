#!/bin/bash
# A/B build of the 1-D beam kernels: scripts/build_b1_variant.sh NAME [extra hipcc flags]  ->  scripts/_timing/libporeover_hip_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p scripts/_timing
python3 poreover_amd/build.py >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -c poreover_amd/csrc/po_beam1d.hip -o /tmp/po_b1_$name.o
objs=$(ls poreover_amd/csrc/_obj/*.o | grep -v po_beam1d)
hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/po_b1_$name.o -o scripts/_timing/libporeover_hip_$name.so
echo scripts/_timing/libporeover_hip_$name.so
