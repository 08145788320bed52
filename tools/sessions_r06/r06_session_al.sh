R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06al; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_arch.py tests/test_gpu_train.py -q -m gpu -rP -k "arch or layerwise or another" > $O/tests.txt 2>&1; echo "tests rc=$?"; grep -E "passed|failed|worst parameter-gradient|Error" $O/tests.txt | cut -c1-250 | tail -20
