"""Aggregate a rocprofv3 kernel trace of bench.py into per-step kernel time (last N optimizer steps).
usage: python scripts/step_profile.py <kernel_trace.csv> [n_steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
opt = [i for i, k in enumerate(names) if 'multi_tensor_apply' in k]
marks = [i for j, i in enumerate(opt) if j + 1 == len(opt) or opt[j + 1] - i > 4]      # last optimizer kernel of every step
lo, hi = marks[-n - 1] + 1, marks[-1] + 1
agg = {}
for r in rows[lo:hi]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(r['Kernel_Name'][:120], [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
span = (int(rows[hi - 1]['End_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e3
print('kernel time %.1f us/step, wall span %.1f us/step' % (tot / n, span / n))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print('%8.1f us/step  x%-5.1f %s' % (v[1] / n, v[0] / n, k))
