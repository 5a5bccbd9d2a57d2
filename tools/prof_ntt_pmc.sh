#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): PMC passes over tools/gpu_ntt_once.py (four 2^24 transforms), one rocprofv3 run per
# counter group; the per-kernel summary lands in gpurun_out/prof_ntt/pmc_per_kernel.json.
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_ntt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o ntt -- python3 $R/tools/gpu_ntt_once.py > $OUT/ntt_stats.log 2>&1 || { echo "kernel trace failed"; exit 1; }
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/pmc_$i -o pmc -- python3 $R/tools/gpu_ntt_once.py > $OUT/pmc_$i.log 2>&1
  rc=$?; echo "pass $i rc=$rc"; tail -2 $OUT/pmc_$i.log
  [ $rc -eq 0 ] || exit 1
done
python3 $R/tools/prof_summarize.py $OUT
