cd $GRAFT_REPO_ROOT
for a in 8 4 2 1; do
for leaf in 96; do
echo "== ALIGN_MIN_BLOCKS=$a LEAF=$leaf"
PANGULU_AMD_ND_ALIGN_MIN_BLOCKS=$a PANGULU_AMD_ND_LEAF=$leaf timeout 600 python bench.py --no-profile-pass --no-cpu-baseline --steps 4 --warmup 2 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['config']; print('ms %.1f' % d['ms_per_step'], 'GF/s %.0f' % d['value'], 'flop %.4g' % c['flop'], 'symbolic nnz %.4g' % c['symbolic_nnz'], 'blocks', c['blocks'], 'tasks', c['tasks'], 'batches', d['batches_per_step'], 'res %.1e' % d['residual'], 'init', d['init_s'])"
done; done
