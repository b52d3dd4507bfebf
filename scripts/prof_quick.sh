#!/bin/bash
# quick kernel trace of the pipelined c2 step (TAG=name): kernel stats + critical path under gpurun_out/$TAG
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG:-quick}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ab $EXTRA > $O/bench_line.json 2> $O/trace.err
echo "trace rc=$?"
ks=$(find $O/trace -name "*kernel_stats.csv" | head -1); kt=$(find $O/trace -name "*kernel_trace.csv" | head -1)
cp $ks $O/kernel_stats.csv
python3 $R/scripts/gpu_timeline.py $kt 10 > $O/critical_path.txt 2>&1
python3 $R/scripts/gpu_gaps.py $kt 10 > $O/gpu_idle_gaps.txt 2>&1
rm -rf $O/trace
