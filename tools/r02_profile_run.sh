set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02p
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02p/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02p/pytest_gpu.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r02p/prof -- python3 bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r02p/bench_under_rocprof.json 2> gpurun_out/r02p/bench_under_rocprof.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c -d gpurun_out/r02p/pmc_$c -- python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --height 544 --width 960 > gpurun_out/r02p/pmc_$c.json 2> gpurun_out/r02p/pmc_$c.err
  HEVCE_CLUSTER=1 timeout -k 10 200 rocprofv3 --pmc $c -d gpurun_out/r02p/pmc1_$c -- python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --height 544 --width 960 > gpurun_out/r02p/pmc1_$c.json 2> gpurun_out/r02p/pmc1_$c.err
done
HEVCE_CLUSTER=1 timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES -d gpurun_out/r02p/pmc1_INST -- python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --height 544 --width 960 > gpurun_out/r02p/pmc1_INST.json 2> gpurun_out/r02p/pmc1_INST.err
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES -d gpurun_out/r02p/pmc_INST -- python3 bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --height 544 --width 960 > gpurun_out/r02p/pmc_INST.json 2> gpurun_out/r02p/pmc_INST.err
find gpurun_out/r02p -name "*.csv" | head -30
