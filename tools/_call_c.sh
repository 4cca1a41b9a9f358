set -u
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r03/ab
timeout 900 python -m pytest tests/test_gpu_mlp.py tests/test_gpu_trainer_h64.py -x -q 2>&1 | tail -4
for w in smac ns_rnn; do
timeout 300 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r03/ab/${w}_nt4.json; cut -c1-250 gpurun_out/r03/ab/${w}_nt4.json
done
