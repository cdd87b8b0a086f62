#!/bin/bash
# A/B runs inside ONE gpurun call: the given command once with the product library and once per build/variants/lib_N.so
# (tools/mkvariant.sh), the variant LOADED through DD_LIB_PATH -- the product .so is never overwritten.
#   tools/each_variant.sh python tools/fft_ab.py          ROUNDS_V=2 repeats the whole sweep
cd "$(dirname "$0")/.."
[ -f build/variants/index.txt ] && cat build/variants/index.txt
for r in $(seq 1 ${ROUNDS_V:-1}); do
  echo "== product library"; "$@"
  for f in build/variants/lib_*.so; do
    [ -f "$f" ] || continue
    i=$(basename $f .so | sed 's/lib_//')
    echo "== variant $(sed -n ${i}p build/variants/index.txt)"
    DD_LIB_PATH=$f "$@"
  done
done
