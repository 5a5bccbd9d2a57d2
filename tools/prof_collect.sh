#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): kernel-trace stats of the shipped bench command and three separate PMC passes
# over one 2^22 MSM process; compact summaries land in gpurun_out/prof/.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $R/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/bench.err || { echo "bench under rocprofv3 failed"; exit 1; }
for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"; do
  tag=$(echo $ctr | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$tag -o pmc -- python3 $R/tools/gpu_one_msm.py 22 0 > $OUT/pmc_$tag.log 2>&1 || { echo "pmc pass $tag failed"; exit 1; }
done
python3 $R/tools/prof_summarize.py $OUT
