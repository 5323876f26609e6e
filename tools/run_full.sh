# full round check on the GPU box: -m gpu tests, smoke, bench (default), step counters -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
mkdir -p gpurun_out/$TAG
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
tail -4 gpurun_out/$TAG/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/$TAG/smoke.log
( time timeout 900 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err ) 2>&1 | grep real; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/$TAG/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','rtf','step_ms'): print(k, d.get(k))
for k in ("audio_only_call","dense_worst_case","moving_f0","single_stream","single_stream_graph","single_stream_native","native_group_call","whole_file"): print(k, {kk:vv for kk,vv in d.get(k,{}).items() if kk!='workload'})
print('roofline', {k:d['roofline'][k] for k in ('frac','ms_per_launch')})
print('roofline_step', d['roofline_step'])
c=d['cpu_baseline']; print('cpu', c['value'], c['cores'], c['kind'], c['sample'][:160])
print('midi_like', {kk:vv for kk,vv in d.get('midi_like',{}).items() if kk!='workload'})
ch=d['roofline'].get('three_operator_chain'); print('chain', ch and ch['ms'], ch and ch['frac_of_measured_write'])
PY
