cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.log; python -c "
import json; d=json.loads(open('gpurun_out/bench_default.log').read()); print('N=1 ms %.1f GF/s %.0f' % (d['ms_per_step'], d['value']), 'frac %.3f' % d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'cpu', d['cpu_baseline']['value'])"
