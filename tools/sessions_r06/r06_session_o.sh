R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06o; mkdir -p $O; cd $R; export TMPDIR=/tmp
python tools/ray_bias_probe.py 2>&1 | tail -1 | tee $O/ray_bias_probe.txt
python tools/ray_bias_probe.py 100000 2>&1 | tail -1 | tee -a $O/ray_bias_probe.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "stages or render or edges or frames" > $O/tests_k.txt 2>&1; echo "tests rc=$?"; tail -3 $O/tests_k.txt | cut -c1-250
