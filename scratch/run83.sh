cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0 PANGULU_AMD_STALL_S=25 PANGULU_TEST_RANK_TIMEOUT=120
for i in 1 2 3; do timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1; done
for i in $(seq 1 8); do
timeout 300 python -m pytest tests/test_multirank.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/mr_dbg_$i.log
if grep -q "7 passed" gpurun_out/mr_dbg_$i.log; then rm gpurun_out/mr_dbg_$i.log; echo "ok $i"; else echo "FAIL $i"; fi
done
for w in "fem27 --size 48" "shell --size 200 200 --nb 128"; do
timeout 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['workload'], 'GF/s %.0f' % d['value'], 'res %.1e' % d['residual'])"
done
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], {k: v['ms'] for k, v in d['kernels'].items() if k == 'getrf'}, 'res %.3e' % d['residual'])"; done
