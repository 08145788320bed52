R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06n; mkdir -p $O; cd $R; export TMPDIR=/tmp
for rep in 1 2; do
python tools/ray_bias_probe.py 2>&1 | tail -1
for v in nostore noload none; do OBJNERF_LIB=$R/object_nerf_amd/tune/libobjnerf_rb_$v.so python tools/ray_bias_probe.py 2>&1 | tail -1; done
done | tee $O/ray_bias_probe.txt
