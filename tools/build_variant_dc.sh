#!/bin/bash
# build_variant_dc.sh NAME [extra hipcc flags...] — A/B builds of the device coder: compiles devcoder.hip with the extra flags and
# links it with the current objects into libbsc_amd/lib/variants/libbsc_NAME.so (load with BSC_LIB_OVERRIDE=<path>).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
python -m libbsc_amd.build > /dev/null
OUT=libbsc_amd/lib/variants; mkdir -p $OUT
hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -I include -I libbsc_amd/csrc --offload-arch=gfx950 "$@" \
      -c libbsc_amd/csrc/device/devcoder.hip -o $OUT/devcoder_$NAME.o 2>/dev/null
OBJS=$(ls libbsc_amd/lib/obj/*.o | grep -v devcoder.o)
hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libbsc_$NAME.so $OBJS $OUT/devcoder_$NAME.o -lpthread -Wl,-Bsymbolic
echo built $OUT/libbsc_$NAME.so
