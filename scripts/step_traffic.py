"""Whole-step HBM traffic from two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE; see profiles/traffic.json's note
for the gfx950 correction): sums the counters over the kernels of ONE WHOLE EPOCH of the timed loop (bench.py walks the cohort: at the
defaults an epoch is 3 steps -- 32 768 + 32 768 + 9 464 encounters) and states them per encounter and per average step.
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/step_fetch -o p -- python3 bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 3 --kernel-iters 1
    rocprofv3 --pmc WRITE_SIZE ... -d gpurun_out/step_write ...
    python scripts/step_traffic.py gpurun_out/step_fetch/p_counter_collection.csv gpurun_out/step_write/p_counter_collection.csv <round> [steps per epoch = 3] [encounters per epoch = 75000]"""
import csv, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import csrc_sha16          # noqa: E402

NB = int(sys.argv[4]) if len(sys.argv) > 4 else 3
N_ENC = int(sys.argv[5]) if len(sys.argv) > 5 else 75000


def per_step(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'adam_amsgrad_kernel' in r['Kernel_Name']]
    # 3 warm-up steps = epoch 0; steps 3..5 = epoch 1 of the run, the first timed one (bench.py's kernel table and step trace launch more kernels afterwards)
    lo, hi = marks[NB - 1] + 1, marks[2 * NB - 1] + 1
    by, kern = {}, {}
    for r in rows[lo:hi]:
        k = r['Kernel_Name']
        c = 'lstm_recurrence' if 'lstm_' in k else 'library_gemm' if 'Cijk' in k else 'other_hip_kernels' if 'dic' in k else 'torch_elementwise'
        by[c] = by.get(c, 0.0) + float(r['Counter_Value']) * 1024 / NB
        short = k.split('(')[0].replace('void ', '')[:64]
        kern[short] = kern.get(short, 0.0) + float(r['Counter_Value']) * 1024 / NB
    return by, kern


(f, fk), (w, wk) = per_step(sys.argv[1], 'FETCH_SIZE'), per_step(sys.argv[2], 'WRITE_SIZE')
out = {'_note': 'HBM bytes per AVERAGE joint step of one epoch of bench.py\'s timed loop (%d steps over %d encounters: full batches + the short last one): '
                'read = 2*FETCH_SIZE*1024 (gfx950 correction), write = WRITE_SIZE*1024; separate rocprofv3 --pmc passes (scripts/step_traffic.py)' % (NB, N_ENC)}
for c in sorted(set(f) | set(w)):
    out[c] = {'read_bytes': int(2 * f.get(c, 0)), 'write_bytes': int(w.get(c, 0))}
per_kernel = {k: int(2 * fk.get(k, 0) + wk.get(k, 0)) for k in set(fk) | set(wk)}
out['per_kernel_bytes(top)'] = dict(sorted(per_kernel.items(), key=lambda kv: -kv[1])[:24])
out['total_bytes'] = int(sum(v['read_bytes'] + v['write_bytes'] for k, v in out.items() if isinstance(v, dict) and 'read_bytes' in v))
out['bytes_per_encounter'] = round(out['total_bytes'] * NB / N_ENC, 1)
out['_batch'], out['_round'], out['_csrc_sha16'] = N_ENC / NB, int(sys.argv[3]) if len(sys.argv) > 3 else 2, csrc_sha16()
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', 'step_traffic.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
