"""ctypes binding of the C ABI in ``include/dic_hip.h`` (``libdic_hip.so``).

This is the *only* compute back end of the package: there is no eager/PyTorch fallback for the
hot-path operators.  If the shared object is missing, or a call returns a non-zero status, a
``RuntimeError`` is raised (C status codes -> Python exceptions).
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
import threading

import torch

from . import switches

switches.validate()          # unknown DIC_* names / values outside the table raise here, once (switches.py)
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DIC_LIB_PATH') or os.path.join(_HERE, 'libdic_hip.so')      # (DIC_LIB_PATH: another build of the library, for A/B runs)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'dic_hip.h')
CSRC_DIR = os.path.join(_HERE, 'csrc')

KM_STATUS_WORDS = 8
DTYPE_F32, DTYPE_BF16, DTYPE_F32X3 = 0, 1, 2
MAX_CHANNELS, MAX_REFPOINTS, MAX_CLUSTERS, LATENT_MAX_DIM = 16, 64, 32, 256

_lib = None
_lock = threading.Lock()

_p, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
# name -> (restype, argtypes); must list every symbol declared in include/dic_hip.h
SIGNATURES = {
    'dic_version': (_i, []),
    'dic_status_string': (C.c_char_p, [_i]),
    'dic_last_error_string': (C.c_char_p, []),
    'dic_sci_cci_fwd': (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    'dic_sci_cci_fwd_ragged': (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    'dic_sci_cci_fwd_store': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    'dic_rbf_fwd_store': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _i, _p, _p, _i, _p, _p, _sz, _p]),
    'dic_rbf_bwd_store': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'dic_sci_cci_bwd_workspace': (_sz, [_i, _i, _i]),
    'dic_sci_cci_bwd': (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _sz, _p]),
    'dic_sci_cci_fwd_packed': (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p]),
    'dic_sci_cci_bwd_packed': (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _sz, _p]),
    'dic_cci_fwd': (_i, [_p, _p, _i, _i, _i, _p, _p]),
    'dic_cci_bwd_workspace': (_sz, [_i, _i, _i]),
    'dic_cci_bwd': (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _sz, _p]),
    'dic_rbf_fwd': (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _i, _p]),
    'dic_rbf_bwd_workspace': (_sz, [_i, _i, _i, _i]),
    'dic_rbf_bwd': (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'dic_rbf_fwd_loss_workspace': (_sz, [_i, _i, _i, _i]),
    'dic_rbf_fwd_loss': (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _i, _p, _p, _sz, _p]),
    'dic_rbf_bwd_loss': (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _sz, _p]),
    'dic_masked_sse_workspace': (_sz, [_i, _i, _i]),
    'dic_masked_sse_fwd': (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _sz, _p]),
    'dic_masked_sse_bwd': (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _i, _p]),
    'dic_dec_fwd_workspace': (_sz, [_i, _i, _i]),
    'dic_dec_fwd': (_i, [_p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _sz, _p]),
    'dic_dec_target': (_i, [_p, _p, _i, _i, _p, _p]),
    'dic_dec_bwd_workspace': (_sz, [_i, _i, _i]),
    'dic_dec_bwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _sz, _p]),
    'dic_dec_kl_workspace': (_sz, [_i, _i]),
    'dic_dec_kl': (_i, [_p, _p, _i, _i, _f, _f, _p, _p, _p, _sz, _p]),
    'dic_kmeans_workspace': (_sz, [_i, _i, _i, _i]),
    'dic_kmeans_lloyd_iter': (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    'dic_kmeans_stats_words': (_sz, [_i, _i]),
    'dic_kmeans_lloyd_partial': (_i, [_p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    'dic_kmeans_lloyd_finish': (_i, [_i, _i, _i, _p, _p, _p, _p]),
    'dic_kmeans_predict': (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    'dic_lstm_fwd': (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    'dic_lstm_fwd_proj': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'dic_lstm_bwd_workspace': (_sz, [_i]),
    'dic_lstm_bwd': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _sz, _i, _i, _p]),
    'dic_lstm_pack': (_i, [_i, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    'dic_lstm_rec_fwd': (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p]),
    'dic_lstm_rec_fwd_proj': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p]),
    'dic_lstm_rec_fwd_proj_x3': (_i, [_p, _p, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p]),
    'dic_lstm_fwd_xproj': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p]),
    'dic_lstm_rec_bwd_workspace': (_sz, [_i]),
    'dic_lstm_rec_bwd': (_i, [_i, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _sz, _i, _i, _p]),
    'dic_lstm_dx_tile_x3': (_i, [_p, C.c_int64, _p, C.c_int64, C.c_int64, _i, _i, _p, _p]),
    'dic_lstm_dw_x3_workspace': (_sz, [_i, _i, _i]),
    'dic_lstm_dw_x3': (_i, [_p, C.c_int64, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _p, _i, _p, _sz, _p]),
    'dic_lstm_dw_workspace': (_sz, [_i, _i]),
    'dic_lstm_dw': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _sz, _p]),
    'dic_lstm_dw_wide_workspace': (_sz, [_i, _i]),
    'dic_lstm_dw_wide': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _sz, _p]),
    'dic_lstm_dx_tile': (_i, [_p, _p, C.c_int64, _i, _i, _p, _p]),
    'dic_lstm_unpack_grads': (_i, [_p, _i, _p, _p, _i, _i, _p, _i, _p]),
    'dic_row_proj': (_i, [_p, _p, _p, C.c_int64, _i, _i, _p, _i, _i, _p]),
    'dic_row_proj_stats_workspace': (_sz, [C.c_int64, _i]),
    'dic_row_proj_stats': (_i, [_p, _p, _p, C.c_int64, _i, _i, _p, _p, _p, _sz, _p]),
    'dic_fc_bwd_workspace': (_sz, [C.c_int64, _i, _i]),
    'dic_fc_bwd': (_i, [_p, _p, _p, C.c_int64, _i, _i, _p, _p, _p, _sz, _p]),
    'dic_fc_bwd_bnhead': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, C.c_float, _p, _p, _p, C.c_int64, _i, _i, _p, _p, _p, _sz, _p]),
    'dic_head_fwd': (_i, [_p, _p, _p, C.c_int64, _i, _i, _p, _p]),
    'dic_head_bwd_workspace': (_sz, [C.c_int64, _i, _i]),
    'dic_head_bwd': (_i, [_p, _p, _p, C.c_int64, _i, _i, _p, _p, _p, _p, _sz, _p]),
    'dic_bn_colstats_workspace': (_sz, [C.c_int64, _i]),
    'dic_bn_colstats': (_i, [_p, C.c_int64, _i, _p, _p, _sz, _p]),
    'dic_bn_moments': (_i, [_p, _i, _f, _f, _p, _p, _p, _p, _p, _p, _p]),
    'dic_bnhead_fwd': (_i, [_p, _p, _p, _p, _p, _p, _p, C.c_int64, _i, _i, _i, _f, _p, _p, _p]),
    'dic_bnhead_bwd_workspace': (_sz, [C.c_int64, _i, _i]),
    'dic_bnhead_bwd_reduce': (_i, [_p, _p, _p, _p, _p, _p, _p, C.c_int64, _i, _i, _i, _f, _p, _p, _p, _sz, _p]),
    'dic_bnhead_bwd_input': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, C.c_double, _p, C.c_int64, _i, _i, _i, _f, _p, _p, _p]),
    'dic_bn_colstats_f32': (_i, [_p, C.c_int64, _i, _p, _p, _sz, _p]),
    'dic_bnhead_fwd_f32': (_i, [_p, _p, _p, _p, _p, _p, _p, C.c_int64, _i, _i, _i, _f, _p, _p, _p]),
    'dic_bnhead_bwd_reduce_f32': (_i, [_p, _p, _p, _p, _p, _p, _p, C.c_int64, _i, _i, _i, _f, _p, _p, _p, _sz, _p]),
    'dic_bnhead_bwd_input_f32': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, C.c_double, _p, C.c_int64, _i, _i, _i, _f, _p, _p, _p]),
    'dic_cluster_pairdist': (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p]),
    'dic_cluster_intra_sums': (_i, [_p, _p, _i, _i, _i, _p, _p]),
    'dic_cluster_intra_totals_workspace': (_sz, [C.c_int64, _i]),
    'dic_cluster_pair_rowsums_workspace': (_sz, [C.c_int64, _i]),
    'dic_cluster_pair_rowsums': (_i, [_p, C.c_long, _p, C.c_int64, _i, _i, _p, _i, _p, _i, _p, _p, _sz, _p]),
    'dic_cluster_intra_totals': (_i, [_p, C.c_long, _p, _p, C.c_int64, _i, _i, _p, _i, _p, _p, _sz, _p]),
    'dic_adam_amsgrad_step': (_i, [_p, _p, _p, _p, _p, C.c_int64, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p]),
    'dic_grad_norm_workspace': (_sz, [C.c_int64]),
    'dic_grad_norm_clip': (_i, [_p, C.c_int64, _f, _p, _p, _sz, _p]),
    'dic_accumulate_many': (_i, [_p, _p, _p, _i, _p]),
    'dic_gemm_nt': (_i, [_i, _i, _p, C.c_int64, _p, C.c_int64, _p, C.c_int64, _i, _i, _p, C.c_int64, _i, _p]),
    'dic_gemm_nt_planes': (_i, [_p, C.c_int64, C.c_int64, _p, C.c_int64, _p, C.c_int64, _i, _i, _p, C.c_int64, _p]),
    'dic_gemm_tn_planes': (_i, [_p, C.c_int64, C.c_int64, _p, C.c_int64, C.c_int64, _i, _i, _p, C.c_int64, _i, _p, C.c_int64, _i, _p, C.c_int64, _i, _i, _p, _sz, _p]),
    'dic_x3_row_proj': (_i, [_p, _p, _p, C.c_int64, _i, _i, _p, _i, _p]),
    'dic_gemm_tn_workspace': (_sz, [C.c_int64, _i, _i, _i]),
    'dic_gemm_tn': (_i, [_i, _p, C.c_int64, _p, C.c_int64, C.c_int64, _i, _i, _p, C.c_int64, _i, _p, C.c_int64, _i, _p, C.c_int64, _i, _i, _p, _sz, _p]),
    'dic_segment_sum_workspace': (_sz, [_i, _i]),
    'dic_segment_sum_f64': (_i, [_p, C.c_int64, _p, _i, _i, _i, _p, _p, _sz, _p]),
    'dic_cumsum_f64': (_i, [_p, _i, _i, _p, _p]),
    'dic_kmeans_pp_workspace': (_sz, [_i, _i]),
    'dic_kmeans_pp_candidates': (_i, [_p, _i, _i, _p, _i, _i, _p, _p, _p, _p, _sz, _p]),
    'dic_kmeans_pp_candidates_rows': (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _sz, _p]),
}


def header_symbols():
    """Every ``dic_*`` function declared in include/dic_hip.h (used by the ABI test)."""
    with open(HEADER_PATH) as f:
        text = re.sub(r'/\*.*?\*/', '', f.read(), flags=re.S)
    return sorted(set(re.findall(r'\b(dic_[a-z0-9_]+)\s*\(', text)))


def build(verbose=False):
    """Compile libdic_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    res = subprocess.run(['make', '-C', CSRC_DIR, '-j4'], capture_output=True, text=True)
    if verbose or res.returncode:
        print(res.stdout[-4000:], res.stderr[-4000:])
    if res.returncode:
        raise RuntimeError('building libdic_hip.so failed')
    return LIB_PATH


def lib():
    """Load (once) and return the ctypes handle.  Fails loudly if the extension is missing."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f'{LIB_PATH} is missing: the HIP extension is the only implementation of the hot path. '
                        'Build it with `python -c "import __graft_entry__ as g; g.build()"` '
                        'or `make -C deep_interpolation_clustering_amd/csrc`.')
                handle = C.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        L = lib()
        raise RuntimeError(f'{what} failed: {L.dic_status_string(status).decode()} '
                           f'({L.dic_last_error_string().decode()})')


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """Host array of device pointers (``const float* const*``); keep the returned object alive for the duration of the call."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def stream_of(t):
    """hipStream_t of torch's current stream on the tensor's device."""
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError('deep_interpolation_clustering_amd: hot-path operators run only on an MI355X '
                               f'(got a {t.device} tensor); there is no CPU fallback by design')


def f32c(t):
    """float32 + contiguous (no copy when already so)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()
