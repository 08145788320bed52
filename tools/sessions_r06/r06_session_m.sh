R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06m; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_arch.py -x -q -m gpu > $O/tests_train.txt 2>&1; echo "train tests rc=$?"; tail -3 $O/tests_train.txt | cut -c1-300
bash tools/train_ab.sh r06m many 4 OBJNERF_WGRAD_BIG_ROUNDS=many 2>&1 | tail -8
