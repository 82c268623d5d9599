#!/bin/bash
# extra PMC passes of the default bench (instruction cache, branches, fetches): pmc_extra.sh <tag>
set -e
TAG=${1:-extra}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS SQ_ACTIVE_INST_MISC" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/bench$i.json 2> $OUT/pmc$i.err || echo "pass $i ($set) failed" >> $OUT/failed.txt
  echo "pass $i done: $set"
done
cd $ROOT
python3 tools/summarise_pmc.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
