set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
echo skip > gpurun_out/r02b/pytest.log
cat gpurun_out/r02b/pytest.log
python benchmarks/bench_run_route.py > gpurun_out/r02b/run_route.json 2>gpurun_out/r02b/rr.err; cat gpurun_out/r02b/run_route.json; tail -3 gpurun_out/r02b/rr.err
