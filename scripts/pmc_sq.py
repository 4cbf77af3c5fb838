"""SQ counter summary per hand-written kernel from rocprofv3 --pmc passes of scripts/kbench.py (one JSON; derived ratios included):
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS \
        --kernel-trace --output-format csv -d /tmp/sq1 -o p -- python3 scripts/kbench.py 32768 3
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE ... -d /tmp/sq2 ...
    python scripts/pmc_sq.py out.json /tmp/sq1/p_counter_collection.csv /tmp/sq2/p_counter_collection.csv
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md); the ratios below are
fractions of wave lifetime."""
import csv, json, sys

acc = {}
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name']
        if 'dic::' not in k:
            continue
        name = k.split('dic::')[1].split('(')[0]
        a = acc.setdefault(name, {}).setdefault(r['Counter_Name'], [0, 0.0])
        a[0] += 1
        a[1] += float(r['Counter_Value'])
out = {}
for name, cs in acc.items():
    d = {c: v[1] / v[0] for c, v in cs.items()}          # mean per launch
    wc = d.get('SQ_WAVE_CYCLES')
    if wc:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU'):
            if c in d:
                d['frac_' + c] = round(d[c] / wc, 4)
    if d.get('SQ_BUSY_CYCLES') and d.get('SQ_ACTIVE_INST_VALU'):
        # VALU issue cycles per SIMD-busy cycle: SQ_BUSY_CYCLES is per SE (32) x cycles; ACTIVE_INST_VALU quad-cycles over all waves
        d['valu_quadcycles_per_busy_cycle'] = round(d['SQ_ACTIVE_INST_VALU'] / d['SQ_BUSY_CYCLES'], 3)
        # share of SIMD cycles with a vector instruction executing: quad-cycles * 4 over (per-SE busy cycles x 1024 SIMDs)
        d['valu_busy_frac_of_simd_cycles'] = round(d['SQ_ACTIVE_INST_VALU'] * 4 * 32 / (d['SQ_BUSY_CYCLES'] * 1024), 3)
    if d.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_bank_conflict_frac'] = round(d.get('SQ_LDS_BANK_CONFLICT', 0.0) / d['SQ_LDS_IDX_ACTIVE'], 4)
    out[name] = {k: (round(v, 1) if isinstance(v, float) and v > 10 else v) for k, v in d.items()}
json.dump(out, open(sys.argv[1], 'w'), indent=1)
for name, d in out.items():
    print(name, {k: v for k, v in d.items() if k.startswith('frac_') or 'per_' in k or 'conflict' in k})
