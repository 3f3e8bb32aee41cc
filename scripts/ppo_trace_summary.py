"""Summarise a rocprofv3 kernel trace of `bench.py --mode ppo --no-graph`: rollout vs update busy time and the
kernel sequence of one optimiser step.  Usage: python scripts/ppo_trace_summary.py <trace dir> [--list]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'vine_step' in r['Kernel_Name']]
it = idx[64:80]
upd = rows[it[-1] + 1: idx[80]]
roll = rows[it[0]:it[-1] + 1]
for rs, label, div in ((roll, 'rollout', 16), (upd, 'update', 32)):
    tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    span = int(rs[-1]['End_Timestamp']) - int(rs[0]['Start_Timestamp'])
    print(label, 'kernels', len(rs), 'busy ms', round(tot / 1e6, 2), 'span ms', round(span / 1e6, 2), 'kernels per step', len(rs) / div)
c, t = collections.Counter(), collections.Counter()
for r in upd:
    n = r['Kernel_Name']
    key = 'GEMM' if n.startswith('Cijk') else n.replace('void at::native::', '').replace('(anonymous namespace)::', '')[:80]
    c[key] += 1
    t[key] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, v in t.most_common(14):
    print('   %7.2f ms %5d  %s' % (v / 1e6, c[k], k))
if '--list' in sys.argv:
    pl = [i for i, r in enumerate(upd) if 'ppo_loss' in r['Kernel_Name']]
    for r in upd[pl[10] + 1:pl[11] + 1]:
        n = r['Kernel_Name']
        n = ('GEMM ' + n.split('_MT')[1][:14]) if n.startswith('Cijk') else n.replace('void at::native::', '').replace('(anonymous namespace)::', '')[:90]
        print('%6.1f us  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, n))
