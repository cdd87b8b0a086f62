bash tools/debug/cold_c4_trace.sh > gpurun_out/cold_trace.txt 2>&1
for r in 1 2 3 4; do
  echo "with code warm-up" >> gpurun_out/cold_trace.txt; python tools/bench_noaa_cold.py /tmp/apt60.npy 2>&1 | grep "^{" >> gpurun_out/cold_trace.txt
  echo "without" >> gpurun_out/cold_trace.txt; DD_NO_CODE_WARMUP=1 python tools/bench_noaa_cold.py /tmp/apt60.npy 2>&1 | grep "^{" >> gpurun_out/cold_trace.txt
done
