cp directdemod_amd/libdirectdemod_hip.so /tmp/lib_orig.so
for f in /tmp/lib_orig.so build/variants/lib_1.so build/variants/lib_2.so build/variants/lib_3.so; do
  cp $f directdemod_amd/libdirectdemod_hip.so
  for w in 1 2 3; do
    echo "$f wgs/cu=$w: $(DD_FFT_WGS_PER_CU=$w KERNELS=fft1k REPS=100 ROUNDS=1 python3 tools/fft_ab.py 2>&1 | grep taps | awk '{print $5, $6}')"
  done
done
cp /tmp/lib_orig.so directdemod_amd/libdirectdemod_hip.so
