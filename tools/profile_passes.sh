#!/bin/bash
# Run on the GPU box from the repo root: kernel-trace stats + PMC passes of the default bench (headline + the
# secondary workloads; separate passes; --pmc never combined with sys/hip/hsa tracing), then the same for the batch path
# (bench.py --workload pendulum_fqi_ann: its own passes, the rollout passes skip it to keep the traces small).
# Output: gpurun_out/<tag>/ summary.txt = per-kernel counter means + the JSON for profiles/pmc_traffic.json (tools/pmc_to_json.py)
set -e
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --no-fqi --no-composite"
FQI="python3 $ROOT/bench.py --no-cpu-baseline --workload pendulum_fqi_ann"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/bench_under_stats.json 2> $OUT/stats.err
echo "stats pass done"
# The batch-path passes launch fqi_epochs_kernel WITHOUT hipLaunchCooperativeKernel (GRLX_FQI_NO_COOP=1: the same kernel on the same grid).
# A process that made a cooperative launch under rocprofv3 ends with SIGSEGV inside ROCR's shut-down (a signal handle whose memory is gone),
# called from the HIP runtime's exit handler -- profiles/r04_fqi_exit_probe.md has the probe that established it.  No exit code is masked here.
export GRLX_FQI_NO_COOP=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fqi_stats -- $FQI > $OUT/fqi_under_stats.json 2> $OUT/fqi_stats.err
echo "fqi stats pass done"
# (counter passes of the rollout workloads: GRLX_ENV_SERVER=0 -- rocprofv3 serialises kernels while it reads counters, and the headline's
#  pair of kernels only exists together; the table accesses of rollout_kernel are the ones of rollout_served_kernel)
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  GRLX_ENV_SERVER=0 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/bench_under_pmc$i.json 2> $OUT/pmc$i.err
  echo "pass $i done: $set"
done
j=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  j=$((j+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/fqi_pmc$j -- $FQI > $OUT/fqi_under_pmc$j.json 2> $OUT/fqi_pmc$j.err
  echo "fqi pass $j done: $set"
done
cd $ROOT
python3 tools/pmc_to_json.py $OUT > $OUT/summary.txt 2>&1 || true
# keep what gets committed (kernel statistics, per-kernel counter means); the raw per-launch csv files are large
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null || true
cp $(find $OUT/fqi_stats -name "*kernel_stats.csv" | head -1) $OUT/fqi_kernel_stats.csv 2>/dev/null || true
rm -rf $OUT/stats $OUT/fqi_stats $OUT/pmc[0-9] $OUT/fqi_pmc[0-9]
cat $OUT/summary.txt
