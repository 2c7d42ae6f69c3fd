#!/bin/bash
# A/B variant of the library without a full rebuild: compile ONE translation unit with extra flags and link it with the other objects of build/obj.
#   tools/ab_build.sh <variant> <unit.hip> [extra hipcc flags...]      -> gpsig_amd/lib/libgpsig_hip_<variant>.so   (run with GPSIG_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
mkdir -p build/ab
obj=build/ab/${name}_$(basename $unit .hip).o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c gpsig_amd/csrc/$unit -o $obj
others=$(ls build/obj/*.o | grep -v "/$(basename $unit .hip).o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpsig_amd/lib/libgpsig_hip_$name.so $obj $others
ls -la gpsig_amd/lib/libgpsig_hip_$name.so
