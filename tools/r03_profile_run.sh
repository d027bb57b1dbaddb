# Round-3 measurement pass on the GPU box (gpurun -- 'bash tools/r03_profile_run.sh'); tools/collect_profiles_r03.py turns gpurun_out/r03p/
# into the tracked summaries under profiles/.  PMC counters in passes of their own (never with a trace), the program directly behind `--`.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03p
mkdir -p $O
( while true; do date +%T >> $O/alive.txt; sleep 60; done ) &
KEEP=$!
B544="bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --height 544 --width 960"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv rocpd -d $O/prof -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c -f csv rocpd -d $O/pmc_$c -- python3 $B544 > $O/pmc_$c.json 2> $O/pmc_$c.err
  HEVCE_CLUSTER=1 timeout -k 10 200 rocprofv3 --pmc $c -f csv rocpd -d $O/pmc1_$c -- python3 $B544 > $O/pmc1_$c.json 2> $O/pmc1_$c.err
done
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES -f csv rocpd -d $O/pmc_INST -- python3 $B544 > $O/pmc_INST.json 2> $O/pmc_INST.err
HEVCE_CLUSTER=1 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES -f csv rocpd -d $O/pmc1_INST -- python3 $B544 > $O/pmc1_INST.json 2> $O/pmc1_INST.err
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv rocpd -d $O/pmc_LDS -- python3 $B544 > $O/pmc_LDS.json 2> $O/pmc_LDS.err
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES -f csv rocpd -d $O/pmc_LDSW -- python3 $B544 > $O/pmc_LDSW.json 2> $O/pmc_LDSW.err
HEVCE_CLUSTER=1 timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv rocpd -d $O/pmc1_LDS -- python3 $B544 > $O/pmc1_LDS.json 2> $O/pmc1_LDS.err
HEVCE_CLUSTER=1 timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES -f csv rocpd -d $O/pmc1_LDSW -- python3 $B544 > $O/pmc1_LDSW.json 2> $O/pmc1_LDSW.err
timeout -k 10 200 python3 $B544 --steps 3 --warmup 1 > $O/bench_960x544.json 2> $O/bench_960x544.err
timeout -k 10 200 python3 tools/gpu_prof.py phases > $O/phases.txt 2>&1
kill $KEEP
find $O -name "*.csv" | head -40
