#!/bin/bash
# round evidence (ROUND=r04 ...): kernel traces (pipelined / per-step), PMC passes.  Run on the GPU box from the repo root.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${ROUND:-r04}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for mode in pipelined unpipelined; do
  extra=""; [ $mode = unpipelined ] && extra="--no-pipeline"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ab $extra > $O/bench_line_under_rocprof_$mode.json 2> $O/trace_$mode.err
  echo "trace $mode rc=$?"
  ks=$(find $O/trace_$mode -name "*kernel_stats.csv" | head -1); kt=$(find $O/trace_$mode -name "*kernel_trace.csv" | head -1)
  cp $ks $O/kernel_stats_$mode.csv
  python3 $R/scripts/gpu_timeline.py $kt 10 > $O/critical_path_$mode.txt 2>&1
  python3 $R/scripts/gpu_gaps.py $kt 10 > $O/gpu_idle_gaps_$mode.txt 2>&1
  rm -rf $O/trace_$mode
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --eager --steps 2 --warmup 1 --no-cpu-baseline --no-ab > /dev/null 2> $O/pmc_$c.err
  echo "pmc $c rc=$?"
  cp $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) $O/pmc_$c.csv
  rm -rf $O/pmc_$c
done
python3 $R/scripts/pmc_hbm_json.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv $O/pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of bench.py --eager --steps 2; bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB"
rm -f $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/pmc_gemm -- python3 $R/scripts/time_gemm.py > $O/time_gemm_under_pmc.txt 2> $O/pmc_gemm.err
echo "pmc gemm rc=$?"
python3 $R/scripts/pmc_summary.py $(find $O/pmc_gemm -name "*counter_collection.csv" | head -1) gemm > $O/pmc_gemm_sq_counters.txt 2>&1
rm -rf $O/pmc_gemm
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/pmc_jacobi -- python3 $R/scripts/time_jacobi.py > $O/time_jacobi_under_pmc.txt 2> $O/pmc_jacobi.err
echo "pmc jacobi rc=$?"
python3 $R/scripts/pmc_summary.py $(find $O/pmc_jacobi -name "*counter_collection.csv" | head -1) jacobi > $O/pmc_jacobi_sq_counters.txt 2>&1
rm -rf $O/pmc_jacobi
python3 $R/scripts/time_gemm.py > $O/time_gemm.txt 2>&1
python3 $R/scripts/time_jacobi.py > $O/time_jacobi.txt 2>&1
python3 $R/scripts/time_procrustes_bwd.py > $O/time_procrustes_bwd.txt 2>&1
python3 $R/scripts/time_token_gram.py > $O/time_token_gram.txt 2>&1
ls -la $O
