"""Per-factorisation kernel timeline from a rocprofv3 kernel trace: python tools/timeline.py <kernel_trace.csv> [first|last N lines]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'rocclr' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
segs = [[rows[0]]]
busy_until = int(rows[0]['End_Timestamp'])
for b in rows[1:]:
    if int(b['Start_Timestamp']) - busy_until > 3e6:
        segs.append([])
    segs[-1].append(b)
    busy_until = max(busy_until, int(b['End_Timestamp']))
s = segs[-1]
t0 = int(s[0]['Start_Timestamp'])
print("factorisations", [len(x) for x in segs], "last: %.2f ms" % ((max(int(r['End_Timestamp']) for r in s) - t0) / 1e6))
short = {'ssssm_dense_f64_kernel': 'SD', 'void ssssm_sparse_kernel<false>': 'SS', 'void trsm_dense_f64_kernel<16>': 'TD', 'trsm_sparse_kernel': 'TS',
         'densify_kernel': 'dn', 'getrf_blocked_f64_kernel': 'G', 'sparsify_kernel': 'sp', 'diag_tile_inverse_kernel': 'di'}
out = []
for r in s:
    k = r['Kernel_Name'].split('(')[0]
    out.append("%-3s %8.2f +%5.0f us  wg %d" % (short.get(k, k), (int(r['Start_Timestamp']) - t0) / 1e6,
               (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) // int(r['Workgroup_Size_X'])))
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (0, len(out))
print("\n".join(out[lo:hi]))
gs = [(int(r['Start_Timestamp']) - t0) / 1e6 for r in s if r['Kernel_Name'].startswith('getrf')]
print("ms between GETRF launches:", [round(b - a, 2) for a, b in zip(gs, gs[1:])])
