#!/bin/bash
# same-call A/B of the default library against every build/variants/lib_N.so: ROUNDS rounds, interleaved, ms per step
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt
for r in $(seq 1 ${ROUNDS:-3}); do
  for f in /tmp/lib_orig.so build/variants/lib_*.so; do
    cp $f directdemod_amd/libdirectdemod_hip.so
    a=$(python bench.py --no-cpu-baseline --no-side 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    b=$(python bench.py --no-cpu-baseline --no-side 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r $f: $a $b"
  done
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
