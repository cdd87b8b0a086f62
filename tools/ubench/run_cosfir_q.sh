#!/bin/bash
# round 6: the running-sum arithmetic with one cosine term (Hamming, three running sums) and three (blackmanHarris, seven), no loads / stores / LDS:
# ms and joules per 2^26 samples, same box -- gate for a four-term product kernel: <= 0.11 ms and <= 0.12 J.   gpurun -- tools/ubench/run_cosfir_q.sh
cd "$(dirname "$0")/../.."
O=gpurun_out/r06_cosfir_q.txt; mkdir -p gpurun_out; : > $O
BIN=tools/ubench/bin/cosfir_arith_q
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o $BIN tools/ubench/cosfir_arith_q.hip
summ() { python3 - "$1" "$2" <<'PY'
import re, sys
sc, pw = [], []
for l in open(sys.argv[1]):
    m = re.search(r'\((\d+)Mhz\),1,\(\d+Mhz\),S,([\d.]+)', l)
    if m:
        sc.append(int(m.group(1))); pw.append(float(m.group(2)))
sc, pw = sc[2:-1], pw[2:-1]
ms = float(sys.argv[2])
if sc:
    P = sum(pw) / len(pw)
    print("   sclk %.0f MHz, package power %.0f W, %d samples: dynamic energy (power - 283 W idle) x time = %.4f J per 2^26 samples" % (sum(sc) / len(sc), P, len(sc), (P - 283.0) * ms * 1e-3))
PY
}
for cfg in "2 1" "2 3" "1 3"; do
  $BIN 5 $cfg > /tmp/cosfir_out.txt 2>&1 &
  pid=$!
  sleep 1.0
  while kill -0 $pid 2>/dev/null; do rocm-smi --showclocks --showpower --csv | tr '\n' ' '; echo; sleep 0.2; done > /tmp/smi.txt
  wait $pid
  cat /tmp/cosfir_out.txt >> $O
  ms=$(grep -o "[0-9.]* ms per 2^26" /tmp/cosfir_out.txt | head -1 | cut -d' ' -f1)
  summ /tmp/smi.txt ${ms:-0} >> $O
done
cat $O
