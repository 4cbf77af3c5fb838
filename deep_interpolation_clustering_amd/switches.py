"""The ``DIC_*`` environment switches: ONE table (name -> allowed values -> default -> effect), validated once when the native binding is
imported (``_native.py``).  None is needed in production -- every fast path is on by default and falls back by itself when its shape
conditions do not hold; the switches exist for A/B measurements and for the tests.  A value outside the allowed set, or an unknown ``DIC_*`` name
that is a near miss of a known one, raises at import (a typo used to be silently the default); any other unknown ``DIC_*`` name -- a site's own
variable, an exported build-macro name -- only warns, and an empty value counts as unset; ``INTEGRATION.md`` section 4 is generated from this table
(``python -m deep_interpolation_clustering_amd.switches --write``).
"""
import os

ON_OFF = ('0', '1')
INT, PATH = 'int', 'path'

# name: (allowed values | INT | PATH, default, where it is read, effect of the non-default value)
SWITCHES = {
    'DIC_ROW_PROJ': (ON_OFF, '1', 'lstm.py', '0: library GEMM for the decoder\'s input projection instead of `dic_row_proj`'),
    'DIC_GX_LANE_NATIVE': (ON_OFF, '1', 'lstm.py', '0: row-major gx between `dic_row_proj` and `dic_lstm_fwd`'),
    'DIC_DEFER_RELU': (ON_OFF, '1', 'lstm.py', '0: the encoder writes a rectified copy of its output instead of the decoder\'s kernels rectifying on load'),
    'DIC_RELU_IN_KERNEL': (ON_OFF, '1', 'lstm.py', '0: `torch.relu` between the LSTMs'),
    'DIC_COMPRESS_FUSED': (ON_OFF, '1', 'ops.py', '0: `CompressFC` as two autograd nodes (`dic_bnhead_bwd_input` + `dic_fc_bwd`)'),
    'DIC_FC_BWD_MIN_ROWS': (INT, '1024', 'ops.py', 'row count from which `CompressFC`\'s first layer uses the resident-weight kernels (8192 until round 3)'),
    'DIC_FWD_EIGHT_WAVES': (ON_OFF, '1', 'lstm.py', '0: four waves per workgroup in the encoder\'s fused-projection recurrence'),
    'DIC_BWD_EIGHT_WAVES': (ON_OFF, '1', 'csrc/dic_lstm.hip', '0: the four-wave backward recurrence `lstm_bwd_kernel` instead of `lstm_bwd8_kernel` (bit-identical; `scripts/bwd8_ab.py`)'),
    'DIC_REC_EIGHT_WAVES': (ON_OFF, '1', 'csrc/dic_lstm32.hip', '0: the four-wave 32-row bf16 recurrence kernels for batches <= 4096 (bit-identical; `scripts/rec8_ab.py`)'),
    'DIC_REC_SIXTEEN': (ON_OFF, '1', 'csrc/dic_lstm32.hip', '0: batches <= 2048 stay on the 32-row eight-wave kernels instead of the 16-row ones (`lstm_rec_fwd16 / bwd16`; bit-identical '
                        'outputs; read per call -- forward and backward of one LSTM call must see the same value)'),
    'DIC_REC16_MAX': (INT, '2048', 'csrc/dic_lstm32.hip', 'largest batch served by the 16-row recurrence kernels'),
    'DIC_LIB_PATH': (PATH, 'in-tree `libdic_hip.so`', '_native.py', 'another build of the library (A/B runs, e.g. one compiled with `-DDIC_NO_NT`)'),
    'DIC_GRAD_SINKS': (ON_OFF, '1', 'ops.py', '0: the small parameter gradients go through autograd\'s AccumulateGrad'),
    'DIC_DW_SIDE_STREAM': (ON_OFF, '0', 'lstm.py', '1: the decoder\'s weight-gradient kernel on a side stream next to the encoder backward (round 2\'s default; round 3 measures '
                           'it 0.03-0.07 ms slower per step: both are bound by the same HBM)'),
    'DIC_SIDE_RECORD_STREAM': (ON_OFF, '0', 'lstm.py', '1: side-stream tensors kept alive through `record_stream` (the allocator-side alternative; experiment)'),
    'DIC_KMEANS_SMALLK_MFMA': (ON_OFF, '1', 'csrc/dic_kmeans_mfma.hip', '0: Lloyd iterations with K <= 8 always on the wave-per-row kernel (round 2), also when several restarts could share X tiles'),
    'DIC_SHARDED_GRAPHS': (ON_OFF, '1', 'step.py', '0: `Stepper(use_graphs=\'auto\')` never captures a SHARDED step (default: on the `nccl` backend the sharded step of a per-rank batch '
                           '<= 8192 is replayed from a hipGraph -- RCCL collectives are stream operations; replay-or-capture is keyed on the GLOBAL batch and a capture counts only '
                           'when every rank succeeded, else all ranks run eagerly; see DESIGN section 6)'),
    'DIC_DIST_BACKEND': (('nccl', 'gloo'), 'nccl on GPUs', 'dist.py', '`gloo`: several ranks may share one GPU (rehearsal of the N > 1 path)'),
    'DIC_DIST_SINGLE_RANK': (ON_OFF, '0', 'dist.py', '1: a process group of ONE rank counts as sharded (every collective runs on the real backend; tests)'),
    'DIC_GEMM_NT8': (ON_OFF, '1', 'csrc/dic_gemm.hip', '0: four waves per 128 x 128 tile in `dic_gemm_nt` (eight: half the work per wave, twice the waves per CU)'),
    'DIC_K1_BWD_LANES': (ON_OFF, '1', 'csrc/dic_interp.hip', '0: the k1 backward stays on the tile kernel at C = 6 (else `sci_cci_bwd_lane_kernel`: a grid point per lane, half a wave per encounter)'),
    'DIC_K1_BWD_LANE_WGS': (INT, '3', 'csrc/dic_interp.hip', 'workgroups per CU of `sci_cci_bwd_lane_kernel`'),
    'DIC_RBF_FWD_ROW': (ON_OFF, '1', 'csrc/dic_rbf.hip', '0: the k2 forward stays on the tile kernel for prefix masks (else `rbf_fwd_row_kernel`: a row per wave; bit-identical)'),
    'DIC_RBF_BWD_SLOT': (('0', '1', '2'), '1', 'csrc/dic_rbf.hip', 'k2 backward: 1 = the slots-on-lanes kernel for every prefix-mask shape the wave-per-encounter kernel (C = 6, R = 24) does '
                         'not take; 2 = for those too; 0 = off (generic tile kernel)'),
    'DIC_SMALL_BATCH': (INT, '4096', 'lstm.py', 'largest batch served by the one-tile-per-workgroup recurrence kernels (`dic_lstm_rec_*`); above it the 64-row kernels and `dic_lstm_fwd_xproj`'),
    'DIC_FWD_XPROJ': (ON_OFF, '1', 'lstm.py', '0: the decoder\'s large-batch forward as `dic_row_proj` + `dic_lstm_fwd` (gx through HBM) instead of `dic_lstm_fwd_xproj`'),
    'DIC_REC_PROJ': (ON_OFF, '1', 'lstm.py', '0: `dic_gemm_nt` + `dic_lstm_rec_fwd` for the encoder\'s forward at batches up to 4096 (gx materialised)'),
    'DIC_X3_DX_TILE': (ON_OFF, '1', 'lstm.py', '0: `dic_gemm_nt_planes` for the decoder\'s input gradient in the x3 step instead of `dic_lstm_dx_tile_x3`'),
    'DIC_X3_DW': (ON_OFF, '1', 'lstm.py', '0: `dic_gemm_tn_planes` (two launches per LSTM) for the x3 step\'s LSTM weight gradients instead of `dic_lstm_dw_x3`'),
    'DIC_X3_REC_PROJ': (ON_OFF, '1', 'lstm.py', '0: `dic_gemm_nt` + a gx tensor for the encoder\'s forward in the x3 step instead of the projection inside `dic_lstm_rec_fwd_proj_x3`'),
    'DIC_X3_ROW_PROJ': (ON_OFF, '1', 'ops.py', '0: `dic_gemm_nt` for the 256-input projections of the x3 step too'),
    'DIC_F32_PRODUCTS': (('exact', 'x3'), 'exact', 'ops.py', 'process-wide default of the f32 step\'s products (what `Stepper(precision=...)` / `--f32_products` select per trainer)'),
    # bench.py / scripts
    'DIC_BENCH_BATCH': (INT, '32768', 'bench.py', '`bench.py --batch` default'),
    'DIC_BENCH_DTYPE': (('bf16', 'f32', 'f32x3'), 'bf16', 'bench.py', '`bench.py --dtype` default'),
    'DIC_BENCH_SCALING': (('weak', 'strong'), 'weak', 'bench.py', '`strong`: one cohort sharded over the ranks, fixed global batch'),
    'DIC_CPU_THREADS': (INT, '16', 'bench.py', 'threads of the CPU baseline'),
    'DIC_CPU_THREADS_MAX': (INT, '64', 'bench.py', 'cap of the all-usable-cores CPU baseline point'),
    'DIC_AB_LIB': (PATH, '-', 'scripts/', 'A/B scripts: path of the second library build'),
    'DIC_FWD8': (ON_OFF, '1', 'scripts/lstm_ab.py', 'A/B scripts: eight-wave forward variant'),
    'DIC_REFERENCE': (PATH, '/root/reference', 'oracle/make_golden*.py', 'fixture generation only: where the upstream repository lies'),
}


def validate(environ=None):
    """Raise ``RuntimeError`` for a value outside a switch's allowed set and for an unknown ``DIC_*`` name that is a near miss of a known switch
    (a typo); warn about any other unknown ``DIC_*`` name (not ours to forbid); an empty value is the switch unset."""
    environ = os.environ if environ is None else environ
    bad, foreign = [], []
    for name, value in environ.items():
        if not name.startswith('DIC_'):
            continue
        spec = SWITCHES.get(name)
        if spec is None:
            import difflib
            near = difflib.get_close_matches(name, SWITCHES, n=1, cutoff=0.85)
            if near:
                bad.append(f'{name}: unknown switch (did you mean {near[0]}?)')
            else:
                foreign.append(name)
            continue
        if value == '':
            continue
        allowed = spec[0]
        if allowed == INT:
            try:
                int(value)
            except ValueError:
                bad.append(f'{name}={value!r}: an integer is expected')
        elif allowed != PATH and value not in allowed:
            bad.append(f'{name}={value!r}: allowed values are {", ".join(allowed)}')
    if foreign:
        import warnings
        warnings.warn('deep_interpolation_clustering_amd: environment variables ' + ', '.join(sorted(foreign)) + ' are not switches of this build (ignored)')
    if bad:
        raise RuntimeError('environment switches of deep_interpolation_clustering_amd: ' + '; '.join(bad) + ' (table: deep_interpolation_clustering_amd/switches.py)')


def get(name, environ=None):
    """The value of a switch (its default when unset), validated."""
    environ = os.environ if environ is None else environ
    if name not in SWITCHES:
        raise KeyError(name)
    validate({name: environ[name]} if name in environ else {})
    return environ.get(name) or SWITCHES[name][1]


BEGIN, END = '<!-- switches:begin (generated from switches.py) -->', '<!-- switches:end -->'


def markdown_table():
    rows = ['| Variable | Allowed | Default | Read in | Effect of the other value |', '|---|---|---|---|---|']
    for name, (allowed, default, where, doc) in SWITCHES.items():
        al = 'integer' if allowed == INT else ('path' if allowed == PATH else ', '.join(f'`{a}`' for a in allowed))
        rows.append(f'| `{name}` | {al} | {default} | `{where}` | {doc} |')
    return '\n'.join(rows)


def write_integration(path=None):
    path = path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'INTEGRATION.md')
    txt = open(path).read()
    a, b = txt.index(BEGIN), txt.index(END)
    open(path, 'w').write(txt[:a] + BEGIN + '\n' + markdown_table() + '\n' + txt[b:])


if __name__ == '__main__':
    import sys
    if '--write' in sys.argv:
        write_integration()
    else:
        print(markdown_table())
