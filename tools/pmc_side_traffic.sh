#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: KiB, separate passes; FETCH_SIZE x 2 on gfx950 for wide coalesced reads, MI355X_MICROARCH.md)
# of the side paths' kernels: the decimating front end (C4 shape, one 2^26-sample chunk) and the kernels of an accurate-sync batch
# (tools/bench_noaa.py 60: 64 windows of 118 152 samples per launch; the last launch of a pass holds 47).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
sed -n '/^cat > \/tmp\/one_decim.py/,/^PY$/p' tools/pmc_decim.sh | sed '1d;$d' > /tmp/one_decim.py
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_sd_$c gpurun_out/pmc_sn_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_sd_$c -o p -- python3 /tmp/one_decim.py > /dev/null 2> gpurun_out/pmc_sd_$c.err
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_sn_$c -o p -- python3 tools/bench_noaa.py 60 > /dev/null 2> gpurun_out/pmc_sn_$c.err
  echo "== $c (KiB per launch, mean over launches)"
  python3 tools/pmc_summary.py gpurun_out/pmc_sd_$c gpurun_out/pmc_sn_$c | grep -A1 -E "k_chain_decim|k_hc_|k_filtfilt_tile|k_filtfilt_cos|k_xcorr_runs_pk|k_scan_final$|k_scan_part"
done
