set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02k
python -m pytest tests/test_gpu_stages.py tests/test_gpu_render.py tests/test_gpu_edges.py -q -m gpu -p no:cacheprovider -x > gpurun_out/r02k/pytest.log 2>&1; echo rc=$?; grep -E "passed|failed" gpurun_out/r02k/pytest.log
cd /tmp; rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02k/trace -o b -- python /root/repo/bench.py --steps 2 --warmup 1 --cpu-rays 0 --split-bf16-steps 0 --pmc off > /root/repo/gpurun_out/r02k/trace.log 2>&1; cd /root/repo
DB=$(ls gpurun_out/r02k/trace/*.db gpurun_out/r02k/trace/*/*.db 2>/dev/null | head -1); python tools/hbm_rates.py $DB
