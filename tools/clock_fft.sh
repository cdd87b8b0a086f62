#!/bin/bash
# per variant library (build/variants/lib_N.so, tools/mkvariant.sh) and the default one: launch time (HIP events, unprofiled) and
# shader cycles per launch (GRBM_GUI_ACTIVE / 8 XCDs, its own --pmc pass) -> the clock the chip held
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
cat build/variants/index.txt
export KERNELS=${KERNELS:-fft1k} ROUNDS=1
for f in /tmp/lib_orig.so build/variants/lib_*.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  t=$(REPS=150 python3 tools/fft_ab.py 2>&1 | grep taps | awk '{print $5}')
  rm -rf gpurun_out/clk
  REPS=20 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/clk -o p -- python3 tools/fft_ab.py > /dev/null 2> gpurun_out/clk.err
  c=$(python3 tools/pmc_summary.py gpurun_out/clk | grep -A2 "k_chain_fft1k" | grep GRBM | sed 's/.*mean=//')
  python3 -c "t=float('$t'); c=float('$c')/8; print('$f  %.4f ms  %.0f cycles/XCD  %.2f GHz' % (t, c, c/(t*1e-3)/1e9))"
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
