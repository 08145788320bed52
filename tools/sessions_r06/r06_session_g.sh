R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R; export TMPDIR=/tmp
PMC_KERNEL=ray_bias_kernel timeout 900 bash tools/pmc_quick.sh > $O/pmc_ray_bias.txt 2>&1; echo "pmc rc=$?"; tail -6 $O/pmc_ray_bias.txt | cut -c1-300
timeout 900 python -m pytest tests -x -q -m gpu -k "two_autograd or ddp_wrapper or training_step_forward or frame or reproducible" > $O/tests_k.txt 2>&1; echo "tests rc=$?"; tail -4 $O/tests_k.txt | cut -c1-250
for rep in 1 2 3; do for nodes in 1 2; do echo "NODES=$nodes"; OBJNERF_TRAIN_NODES=$nodes timeout 300 python tools/train_bench.py 2>&1 | tail -1 | cut -c1-200; done; done | tee $O/train_nodes_ab.txt
