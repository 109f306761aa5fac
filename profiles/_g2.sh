set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r02b/pytest.log 2>&1
cat gpurun_out/r02b/pytest.log
python bench.py > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err; tail -2 gpurun_out/r02b/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02b/bench.json').read().strip().split('\n')[-1])
print('value',d['value'],'ms/step',d['ms_per_step'],'roofline',d['roofline']['frac'],d['roofline']['avg_launch_ms'])
print(d['breakdown_ms'])
print('1k',d['shape_10Mx1k']['ms_per_step'],d['shape_10Mx1k']['breakdown_ms'])
pg=d['permuted_growth']; print('pg',pg['seconds_per_call'],pg['presence_pack_ms'],pg['presence_pack_cover_kernel_ms'])
print(d['cpu_baseline'].get('agrees_with_gpu'), d['cpu_baseline'].get('value'))
PY
