set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -25 ) > gpurun_out/r02b/pytest.log 2>&1
cat gpurun_out/r02b/pytest.log
for ARGS in "" "--lanes 2" "--paths 1024 --steps 40"; do
python bench.py --no-cpu-baseline --no-permuted-growth --no-shape-1k $ARGS 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); b=d['breakdown_ms']
print('ARGS [$ARGS] value %.0f ms/step %.4f cover %.4f idx %.4f hist %.4f host_growth %.3f lat %.3f' % (d['value'], d['ms_per_step'], b['tile_cover'], b['tile_index'], b['hist'], b['host_closed_form_growth'], b['single_pass_latency']))"
done
