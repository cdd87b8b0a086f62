#!/bin/bash
# decimating kernels: non-temporal tile loads / cyclic tile map, A/B in one call (one-chunk passes of the C4 and C3 front ends, and bench.py's side lines)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out/r04_decim_map.txt
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt > $O
for r in 1 2; do
for f in /tmp/lib_orig.so build/variants/lib_1.so build/variants/lib_2.so build/variants/lib_3.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  echo "== $f: $(python3 tools/bench_configs.py 2>/dev/null | head -2 | cut -c1-120 | tr '\n' '|')" >> $O
  echo "   u8: $(python3 tools/bench_u8.py 2>/dev/null | head -1 | cut -c1-120)" >> $O
done
done
cp build/variants/lib_3.so directdemod_amd/libdirectdemod_hip.so
echo "== lib_3 parity: $(timeout 900 python3 -m pytest tests/test_gpu_determinism.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -1)" >> $O
echo "== lib_3 bench side: $(python3 bench.py --no-cpu-baseline --steps 50 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print([(s.get('ms_per_pass'), s.get('one_chunk',{}).get('ms')) for s in j['extra']['side'][:2]])")" >> $O
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
echo "== default bench side: $(python3 bench.py --no-cpu-baseline --steps 50 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print([(s.get('ms_per_pass'), s.get('one_chunk',{}).get('ms')) for s in j['extra']['side'][:2]])")" >> $O
cat $O
