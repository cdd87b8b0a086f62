#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (KiB; FETCH doubled per MI355X_MICROARCH.md) of the FFT kernels, separate passes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
export KERNELS=${KERNELS:-fft1k} REPS=${REPS:-10} ROUNDS=1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_t$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_t$c -o p -- python3 tools/fft_ab.py > /dev/null 2> gpurun_out/pmc_t$c.err
  python3 tools/pmc_summary.py gpurun_out/pmc_t$c | grep -A2 "${PMC_KERNEL:-k_chain_fft}" || tail -3 gpurun_out/pmc_t$c.err
done
