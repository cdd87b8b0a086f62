#!/bin/bash
# HBM traffic of the decimating front ends over one 2^26-sample chunk (CASE = C4 C3 C4u8, default all three): FETCH_SIZE / WRITE_SIZE in
# separate passes (KiB; FETCH_SIZE x 2 on gfx950 for 16-byte-per-lane reads), written into profiles/hbm_traffic.json by tools/decim_traffic_json.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
sed -n '/^cat > \/tmp\/one_decim.py/,/^PY$/p' tools/pmc_decim.sh | sed '1d;$d' > /tmp/one_decim.py
for CASE in ${CASES:-C4 C3 C4u8}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_dw_${CASE}_$c
    CASE=$CASE rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_dw_${CASE}_$c -o p -- python3 /tmp/one_decim.py > /dev/null 2> gpurun_out/pmc_dw_${CASE}_$c.err
    echo "== $CASE $c (KiB per launch, mean over launches)"
    python3 tools/pmc_summary.py gpurun_out/pmc_dw_${CASE}_$c | grep -A1 "k_chain_decim"
  done
done
python3 tools/decim_traffic_json.py
