"""profiles/traffic.json from two rocprofv3 PMC passes of scripts/kbench.py (FETCH_SIZE and WRITE_SIZE collected separately):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o p -- python3 scripts/kbench.py 32768 3
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o p -- python3 scripts/kbench.py 32768 3
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch/p_counter_collection.csv gpurun_out/pmc_write/p_counter_collection.csv 32768
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and on gfx950 FETCH_SIZE reports half of
a wide coalesced read stream (MI355X_MICROARCH.md, HBM section)."""
import csv, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from bench import csrc_sha16          # noqa: E402  (the kernel sources these counters were taken on: bench.py prints traffic only for the same sources)
KEYS = {'sci_cci_fwd_kernel': 'sci_cci_fwd', 'sci_cci_bwd_kernel': 'sci_cci_bwd', 'sci_cci_bwd_lane_kernel': 'sci_cci_bwd', 'rbf_fwd_kernel': 'rbf_fwd', 'rbf_fwd_row_kernel': 'rbf_fwd',
        'rbf_bwd_kernel': 'rbf_bwd', 'rbf_bwd_slot_kernel': 'rbf_bwd',
        'masked_sse_kernel': 'masked_sse_fwd', 'masked_sse_bwd_kernel': 'masked_sse_bwd', 'dec_fwd_kernel': 'dec_fwd',
        'dec_bwd_kernel': 'dec_bwd', 'lstm_fwd_kernel<0>': 'lstm_fwd', 'lstm_fwd_kernel<2>': 'lstm_fwd', 'lstm_fwd8_gxn_kernel': 'lstm_fwd', 'lstm_fwd_kernel<1>': 'lstm_fwd_proj', 'lstm_fwd8_proj_kernel': 'lstm_fwd_proj', 'lstm_fwdx8_kernel': 'lstm_fwd_xproj',
        'lstm_bwd_kernel': 'lstm_bwd', 'lstm_bwd8_kernel': 'lstm_bwd', 'rbf_bwd_wave_kernel': 'rbf_bwd', 'lstm_dw_kernel': 'lstm_dw', 'lstm_dw_wide_kernel': 'lstm_dw_wide',
        'row_proj_kernel<8': 'row_proj', 'row_proj_kernel<4': 'row_proj_stats', 'fc_bwd_kernel': 'fc_bwd', 'dx_tile_kernel': 'lstm_dx_tile'}


def match(kernel_name, pat):
    base = 'dic::' + pat
    if '<' in pat:
        return base in kernel_name
    return (base + '(') in kernel_name or (base + '<') in kernel_name


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        for pat, key in KEYS.items():
            if match(r['Kernel_Name'], pat):
                a = acc.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
                break
    return {k: v[1] / v[0] for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {'_note': 'HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/kbench.py %s 3; see '
                'scripts/pmc_traffic.py): bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; the factor 2 is the gfx950 correction of '
                'MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read stream). bench.py scales it linearly with the batch.' % sys.argv[3],
       '_batch': int(sys.argv[3]), '_round': int(sys.argv[4]) if len(sys.argv) > 4 else 2, '_csrc_sha16': csrc_sha16()}
for k in dict.fromkeys(KEYS.values()):
    if k in fetch and k in write:
        out[k] = {'fetch_size_kb': round(fetch[k], 1), 'write_size_kb': round(write[k], 1), 'hbm_bytes': int((2 * fetch[k] + write[k]) * 1024)}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', 'traffic.json'), 'w'), indent=1)
for k, v in out.items():
    if not k.startswith('_'):
        print('%-16s %8.1f MB' % (k, v['hbm_bytes'] / 1e6))
