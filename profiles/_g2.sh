set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r02b/pytest.log 2>&1
cat gpurun_out/r02b/pytest.log
