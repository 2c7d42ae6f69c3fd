#!/bin/bash
# A/B variant of the library without a full rebuild: compile some translation units with extra flags and link them with the other objects of build/obj.
#   tools/ab_build.sh <variant> <unit.hip[,unit2.hip,...]> [extra hipcc flags...]      -> gpsig_amd/lib/libgpsig_hip_<variant>.so   (run with GPSIG_LIB=...)
set -e
cd "$(dirname "$0")/.."
name=$1; units=$2; shift 2
mkdir -p build/ab
objs=""; skip=""
for unit in ${units//,/ }; do
  b=$(basename $unit .hip)
  obj=build/ab/${name}_$b.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-compress "$@" -c gpsig_amd/csrc/$unit -o $obj &
  objs="$objs $obj"; skip="$skip -e /$b.o\$"
done
wait
others=$(ls build/obj/*.o | grep -v $skip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpsig_amd/lib/libgpsig_hip_$name.so $objs $others
ls -la gpsig_amd/lib/libgpsig_hip_$name.so
