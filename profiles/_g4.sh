set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
rm -f gpurun_out/r02e/growth.json
for W in 12288 24576; do
for R in 128 16; do
PNX_GROWTH_WGS=$W python benchmarks/bench_ordered_growth.py --reps 5 --warm-full --orders $R 2> gpurun_out/r02e/growth.err | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('WGS $W R $R call_ms %.3f kernels_ms %.3f' % (d['seconds_per_call']*1e3, d['growth_kernels_ms_per_call_rank0']))" >> gpurun_out/r02e/growth.json
done; done
cat gpurun_out/r02e/growth.json
