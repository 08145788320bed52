#!/bin/bash
# Round-3: the shipped library (clock probe removed) once more through the training tests; PMC view of the training kernels after
# the weight-gradient rewrite
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03zy; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_abi_and_host.py -x -q -m gpu > $O/test_train.txt 2>&1; echo "train tests rc=$?"; tail -2 $O/test_train.txt
timeout 200 python tools/train_bench.py > $O/train_bench.txt 2>&1; tail -2 $O/train_bench.txt
bash tools/train_pmc.sh $O/train_pmc.md > $O/pmc_log.txt 2>&1; grep "rc=" $O/pmc_log.txt; head -9 $O/train_pmc.md | cut -c1-260
