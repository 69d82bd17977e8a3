#!/bin/bash
# kernel trace of the training bench -> gpurun_out/tr (rocpd database)
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/tr && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/tr -o tr -- python tools/bench_configs.py --what ${1:-train} > gpurun_out/tr/out.log 2>&1
