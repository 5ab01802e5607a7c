#!/bin/bash
# build_variant.sh NAME [extra hipcc flags...] — A/B builds of the radix engine: compiles radix_sort.hip (or SRC=<name> from
# csrc/device, e.g. SRC=radix_onesweep) with the extra flags and links it with the current objects into
# libbsc_amd/lib/variants/libbsc_NAME.so (load with BSC_LIB_OVERRIDE=<path>).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
python -m libbsc_amd.build > /dev/null
OUT=libbsc_amd/lib/variants; mkdir -p $OUT
hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -I include -I libbsc_amd/csrc --offload-arch=gfx950 -DBSC_EXPERIMENT_BUILD "$@" \
      -c libbsc_amd/csrc/device/${SRC:-radix_sort}.hip -o $OUT/${SRC:-radix_sort}_$NAME.o
OBJS=$(ls libbsc_amd/lib/obj/*.o | grep -v "/${SRC:-radix_sort}.o")
hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libbsc_$NAME.so $OBJS $OUT/${SRC:-radix_sort}_$NAME.o
echo built $OUT/libbsc_$NAME.so
