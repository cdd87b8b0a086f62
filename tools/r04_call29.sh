#!/bin/bash
# round 4, GPU call 29: k_xcorr_runs_pk with its workgroups per CU throttled (unused dynamic LDS): does an L2-resident working set pay?
cd ${GRAFT_REPO_ROOT:-/root/repo}
for kb in 0 20 27 32 40 53; do
  echo "== DD_XC_PAD_KB=$kb"
  DD_XC_PAD_KB=$kb tools/noaa_timeline.sh 60 2>&1 | grep -E "k_xcorr_runs_pk|span" | tail -5
done
