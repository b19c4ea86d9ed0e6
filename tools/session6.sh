#!/bin/bash
# round-3 final evidence session: full GPU suite, smoke, headline profile set (r03), bf16 configuration profile sets
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s6; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
bash tools/gpu_profile_session.sh r03 > /dev/null 2>&1
bash tools/gpu_profile_session.sh r03 cmu_v8_bf16_l2 "--precision bf16 --views 8 --depth 2" > /dev/null 2>&1
bash tools/gpu_profile_session.sh r03 cmu_v8_bf16_l12 "--precision bf16 --views 8 --depth 12" > /dev/null 2>&1
tail -c 600 gpurun_out/prof_r03/bench.json; echo; tail -c 400 gpurun_out/prof_r03_cmu_v8_bf16_l2/bench.json
