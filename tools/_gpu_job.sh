timeout -k 10 600 python -m pytest tests/test_dd_gpu.py tests/test_cli_gpu.py -x -q > gpurun_out/r2_dd_tests9.log 2>&1; tail -5 gpurun_out/r2_dd_tests9.log
grep -q "Memory access fault" gpurun_out/r2_dd_tests9.log && exit 3
( time python bench.py ) > gpurun_out/r2_bench9.json 2> gpurun_out/r2_bench9.err; tail -5 gpurun_out/r2_bench9.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench9.json').read().strip().split("\n")[-1])
if 'end_to_end' in d and 'cold_cli' in d['end_to_end']: pass
for k in ('value','ms_per_step','verified_pairs'): print(k, d[k])
print('roofline', d['roofline'])
print('cpu', json.dumps(d['cpu_baseline'])[:900])
e=d.get('end_to_end',{}); print('e2e', {k:v for k,v in e.items() if k!='stdout'})
print('dd', d.get('dd_forced_iterations'))
PY
