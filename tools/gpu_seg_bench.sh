#!/bin/bash
# Runs ON THE GPU BOX: the bench line (no extras, no CPU leg) for several accumulate segment lengths (DVN_MSM_SEG).
set -u
R=$GRAFT_REPO_ROOT
for seg in ${@:-0 64 96 128}; do
  echo "== DVN_MSM_SEG=$seg"
  DVN_MSM_SEG=$seg timeout -k 10 120 python3 $R/bench.py --no-cpu-baseline --no-extras | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'), d['config'].get('depth1_ms_per_msm'), d['config'].get('kernel_ms_solo'))" || exit 1
done
