#!/bin/bash
# round-3 GPU session 1: hardware probe for the h2 engine, GPU test-suite, headline bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/s1; mkdir -p $O; export TMPDIR=/tmp; cd $R
timeout 300 build_tmp/h2_probe > $O/probe.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
timeout 600 python bench.py > $O/bench.log 2>&1
tail -3 $O/tests.log; cat $O/probe.log; tail -c 1500 $O/bench.log
