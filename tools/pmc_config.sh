#!/bin/bash
# PMC passes of one configuration (see tools/run_config.py): pmc_config.sh <tag> <name> <replicas> <warm> <trials>
set -e
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $ROOT/tools/run_config.py "$@" > $OUT/run$i.txt 2> $OUT/pmc$i.err || echo "pass $i failed" >> $OUT/failed.txt
done
cd $ROOT
python3 tools/summarise_pmc.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/run1.txt $OUT/summary.txt
