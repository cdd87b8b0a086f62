#!/bin/bash
# build/variants/lib_<N>.so from the current tree with extra -D flags for one translation unit:
#   mkvariant.sh <N> <unit> [flags...]      e.g.  mkvariant.sh 1 dd_mfma -DDD_WS_SADDR
cd "$(dirname "$0")/.."
N=$1; U=$2; shift 2
mkdir -p build/variants
EXTRA=""; [ "$U" = dd_mfma ] && EXTRA="-fno-slp-vectorize"; [ "$U" = dd_afsk ] && EXTRA="-ffp-contract=off"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -w $EXTRA "$@" -c directdemod_amd/csrc/$U.hip -o build/variants/v$N.o || exit 1
OBJS=""
for u in dd_runtime dd_chain dd_fir dd_mfma dd_fftfir dd_cosfir dd_decimw dd_audio dd_afsk; do
  if [ $u = $U ]; then OBJS="$OBJS build/variants/v$N.o"; else OBJS="$OBJS build/obj/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_$N.so $OBJS -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib && rm build/variants/v$N.o
echo "$N: $U $*" >> build/variants/index.txt
