R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R; export TMPDIR=/tmp
# A/B: embedding gradients folded into the chain (default) vs the two GEMMs (OBJNERF_BWD_DX=0), alternating
bash tools/train_ab.sh r06j dx0 4 OBJNERF_BWD_DX=0 2>&1 | tail -8
python bench.py --steps 2 --warmup 1 --cpu-rays 0 --pmc off --train-steps 20 2>/dev/null | tail -1 > $O/bench_line.json; python -c "
import json; d=json.load(open('$O/bench_line.json')); print(json.dumps(d['train_step'])[:1500])"
PMC_KERNEL=ray_bias_kernel timeout 900 bash tools/pmc_quick.sh > $O/pmc_ray_bias.txt 2>&1; tail -4 $O/pmc_ray_bias.txt | cut -c1-300
