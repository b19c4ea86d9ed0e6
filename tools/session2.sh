#!/bin/bash
# round-3 GPU session 2: first run of the h2 engine
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s2; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_h2_gpu.py -x -q -s > $O/h2.log 2>&1; echo "rc=$?" >> $O/h2.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "golden or launch_modes or independent or full_size" > $O/par.log 2>&1; echo "rc=$?" >> $O/par.log
timeout 600 python bench.py --no-extra --no-cpu-baseline > $O/bench.log 2>&1
timeout 600 python bench.py --no-extra --no-cpu-baseline --precision fp32x3 > $O/bench_x3.log 2>&1
tail -40 $O/h2.log; tail -15 $O/par.log; tail -c 2500 $O/bench.log; echo; tail -c 600 $O/bench_x3.log
