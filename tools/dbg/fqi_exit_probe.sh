#!/bin/bash
# Run on the GPU box from the repo root: each variant of tools/dbg/fqi_exit_probe.py ONCE under `rocprofv3 --kernel-trace --stats`
# (and the failing bench command itself), exit codes, stack traces and the process maps kept under gpurun_out/<tag>/.
TAG=${1:-r04_exit}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in rollout fqi fqi_nocoop fqi_leak; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -- python3 $ROOT/tools/dbg/fqi_exit_probe.py $v $OUT/maps_$v.txt > $OUT/$v.out 2> $OUT/$v.err
  echo "$v: rc $?" | tee -a $OUT/rc.txt
  rm -rf $OUT/prof_$v
done
# without the profiler: does the plain process exit cleanly?
timeout -k 10 300 python3 $ROOT/tools/dbg/fqi_exit_probe.py fqi $OUT/maps_fqi_noprof.txt > $OUT/fqi_noprof.out 2> $OUT/fqi_noprof.err
echo "fqi without rocprofv3: rc $?" | tee -a $OUT/rc.txt
# the failing command of tools/profile_passes.sh itself, with the maps of that process
GRLX_DUMP_MAPS=$OUT/maps_bench.txt timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $ROOT/bench.py --no-cpu-baseline --workload pendulum_fqi_ann > $OUT/bench.out 2> $OUT/bench.err
echo "bench fqi under rocprofv3: rc $?" | tee -a $OUT/rc.txt
rm -rf $OUT/prof_bench
for f in $OUT/*.err; do echo "== $f"; grep -A24 "SIGSEGV\|Aborted at" $f | head -40; done > $OUT/traces.txt
ls /opt/rocm/lib/librocprofiler-sdk* /opt/rocm/lib/libamdhip64* /opt/rocm/lib/libhsa-runtime64* > $OUT/libs.txt 2>&1
cat $OUT/rc.txt
