R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06af; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_arch.py -q -m gpu > $O/tests_arch.txt 2>&1; echo "arch tests rc=$?"; tail -15 $O/tests_arch.txt | cut -c1-300
