#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel trace of tools/gpu_groth16.py 20 and the last proof's timeline
# (tools/prof_groth16_timeline.py) into gpurun_out/prof_g16/timeline.txt.  Environment of the caller is passed on.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_g16${1:-}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o g16 -- python3 $R/tools/gpu_groth16.py 20 > $OUT/run.log 2>&1 || { echo failed; tail -5 $OUT/run.log; exit 1; }
python3 $R/tools/prof_groth16_timeline.py $(dirname $(find $OUT/tr -name '*kernel_trace.csv' | head -1)) > $OUT/timeline.txt
grep groth16 $OUT/run.log
