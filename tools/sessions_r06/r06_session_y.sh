R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06y; mkdir -p $O; cd $R; export TMPDIR=/tmp
timeout 200 python tools/power_probe.py render 12 > $O/power_render.txt 2>&1; tail -6 $O/power_render.txt | cut -c1-250
timeout 200 python tools/power_probe.py train 12 > $O/power_train.txt 2>&1; tail -6 $O/power_train.txt | cut -c1-250
