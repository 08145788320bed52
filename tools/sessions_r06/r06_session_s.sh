R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06s; mkdir -p $O; cd $R; export TMPDIR=/tmp
bash tools/train_ab.sh r06s fx 4 2>&1 | tail -8
OBJNERF_LIB=$R/object_nerf_amd/tune/libobjnerf_fx.so timeout 600 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "reproducible or stream_k or gradients_match" 2>&1 | tail -2
