#!/bin/bash
# arithmetic-only time, clock, package power and joules per 2^26 samples of the cosine-series running-sum
# form (tools/ubench/cosfir_arith.hip) beside the overlap-save FFT kernel's builds on the SAME box:
#   gpurun -- tools/ubench/run_cosfir.sh     (writes gpurun_out/r05_cosfir.txt)
cd "$(dirname "$0")/../.."
O=gpurun_out/r05_cosfir.txt; mkdir -p gpurun_out; : > $O
BIN=tools/ubench/bin/cosfir_arith
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN tools/ubench/cosfir_arith.hip
summ() { python3 - "$1" <<'PY'
import re, sys
sc, pw = [], []
for l in open(sys.argv[1]):
    m = re.search(r'\((\d+)Mhz\),1,\(\d+Mhz\),S,([\d.]+)', l)
    if m:
        sc.append(int(m.group(1))); pw.append(float(m.group(2)))
sc, pw = sc[2:-1], pw[2:-1]
if sc:
    print("   sclk %.0f MHz (min %d max %d), package power %.0f W (min %.0f max %.0f), %d samples" % (sum(sc) / len(sc), min(sc), max(sc), sum(pw) / len(pw), min(pw), max(pw), len(sc)))
PY
}
echo "# idle" >> $O
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower --csv | tr '\n' ' '; echo; sleep 0.2; done > /tmp/smi_idle.txt; tail -1 /tmp/smi_idle.txt | cut -c1-300 >> $O
for cfg in "2 1" "2 0" "1 1"; do
  $BIN 5 $cfg > /tmp/cosfir_out.txt 2>&1 &
  pid=$!
  sleep 1.0
  while kill -0 $pid 2>/dev/null; do rocm-smi --showclocks --showpower --csv | tr '\n' ' '; echo; sleep 0.2; done > /tmp/smi.txt
  wait $pid
  cat /tmp/cosfir_out.txt >> $O
  summ /tmp/smi.txt >> $O
done
echo "# the overlap-save FFT kernel on this box: product, then build/variants (tools/mkvariant.sh)" >> $O
[ -n "$SKIP_FFT" ] || bash tools/debug/clock_power_sweep.sh 2>&1 | grep -v amdgpu.ids >> $O
cat $O
