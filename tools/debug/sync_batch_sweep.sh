for b in 32 64 96 120 239; do echo "DD_SYNC_BATCH=$b"; DD_SYNC_BATCH=$b python tools/bench_noaa.py 60 2>&1 | grep "resident in HBM"; done
