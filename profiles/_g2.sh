set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02b/pytest.log
python benchmarks/bench_tile_index.py --coarse 4,8 --probe 16 --tile-blocks 1,2 > gpurun_out/r02b/k0.jsonl 2> gpurun_out/r02b/k0.err
cat gpurun_out/r02b/pytest.log; cut -c1-150 gpurun_out/r02b/k0.jsonl; tail -3 gpurun_out/r02b/k0.err
