#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace stats of ONE stream of 2^22 G1 MSMs (tools/gpu_one_msm.py), so that every
# kernel's duration is its own (no other lane shares the CUs).  Summary lands in gpurun_out/prof_solo/.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_solo; mkdir -p $OUT
LOGN=${1:-22}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solo -o solo -- python3 $R/tools/gpu_one_msm.py $LOGN 0 > $OUT/solo.log 2>&1 || { echo "solo trace failed"; tail -5 $OUT/solo.log; exit 1; }
cat $OUT/solo.log | tail -2
f=$(find $OUT/solo -name '*kernel_stats.csv' | head -1)
cp $f $OUT/solo_kernel_stats.csv
cut -d, -f1-4,6,7 $OUT/solo_kernel_stats.csv | head -30
