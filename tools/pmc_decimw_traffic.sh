#!/bin/bash
# HBM traffic of k_chain_decim_w (C4 front-end shape, one 2^26-sample complex64 chunk): FETCH_SIZE / WRITE_SIZE, separate passes (KiB; FETCH_SIZE x 2 on gfx950)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
sed -n '/^cat > \/tmp\/one_decim.py/,/^PY$/p' tools/pmc_decim.sh | sed '1d;$d' > /tmp/one_decim.py
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_dw_$c
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_dw_$c -o p -- python3 /tmp/one_decim.py > /dev/null 2> gpurun_out/pmc_dw_$c.err
  echo "== $c (KiB per launch, mean over launches)"
  python3 tools/pmc_summary.py gpurun_out/pmc_dw_$c | grep -A1 "k_chain_decim"
done
