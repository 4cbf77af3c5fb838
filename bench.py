#!/usr/bin/env python3
"""Headline benchmark: encounters/sec through one joint interp+DEC training step (BASELINE.json).

    python bench.py --gpus 1 --steps 30 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 75 000 synthetic encounters, 6 vitals, ~50 irregular samples per
channel over 24 h (T=96 padded slots), R=24 reference points, K=4 clusters.  The cohort is resident in
HBM before the timed region; one step = zero_grad -> Net.forward (HIP interp, bi-LSTM enc/dec, HIP
de-interp, HIP DEC) -> ae_mse + 10*kl -> backward -> RCCL gradient all-reduce -> clip 15 -> Adam(amsgrad)
on a per-GPU batch of --batch encounters (weak scaling: the per-GPU batch is fixed as N grows).

Besides the contract line it reports, for the dominant hand-written kernel, achieved algorithmic HBM GB/s
(HIP-event timed on the launch stream) against the 8 TB/s peak, a per-kernel table, and the CPU oracle
("port" of the reference path, oracle/dic_oracle.py) timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
C, T, H, LAM, R, D = 6, 96, 24.0, 50.0, 24, 256


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=100, help='timed steps (SURVEY.md 8d: >= 100)')
    p.add_argument('--warmup', type=int, default=20, help='untimed warm-up steps (SURVEY.md 8d: >= 20)')
    p.add_argument('--batch', type=int, default=int(os.environ.get('DIC_BENCH_BATCH', 32768)), help='encounters per GPU per step')
    p.add_argument('--encounters', type=int, default=75000, help='cohort size resident per GPU')
    p.add_argument('--clusters', type=int, default=None, help='K (default 4; 8 for the 8-GPU config)')
    p.add_argument('--dtype', choices=['bf16', 'f32', 'f32x3'], default=os.environ.get('DIC_BENCH_DTYPE', 'bf16'),
                   help='bf16: autocast for the bi-LSTMs / FC heads (HIP kernels stay f32); f32: every tensor and product f32 (exact-f32 MFMA recurrence, '
                        'f32 library GEMMs); f32x3: every tensor f32, every dense product a three-term bf16 split on the matrix cores (no library GEMM)')
    p.add_argument('--graph', action='store_true', help='capture the step in a hipGraph (pays off for small --batch)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--fake-detection', action='store_true',
                   help="upstream's default unsupervised loss ae_mse_fake_detect_kl (p3:78 minus the private supervised labels): "
                        "sci/cci/encoder run a second time on corrupted samples; NOT the headline configuration")
    p.add_argument('--dropout', type=float, default=0.0, help='CompressFC / head dropout (upstream default 0.2; the headline uses 0)')
    p.add_argument('--cpu-seconds', type=float, default=15.0)
    p.add_argument('--no-secondary', action='store_true', help='skip the secondary records (cfg4, cfg5, batch256, f32, loss deviation)')
    p.add_argument('--no-sweep', action='store_true', help='skip the p2 K=2..20 sweep inside the cfg5 record')
    p.add_argument('--kernel-iters', type=int, default=5, help='stand-alone launches per kernel of the kernel table (kept few: a rocprofv3 '
                   'average over this command should be dominated by the launches of the timed steps)')
    p.add_argument('--dense-input', action='store_true', help='feed the step padded (B,4C,T) batches instead of reading the ragged encounter '
                   'store in place (A/B of the input path; the default is what the trainers run on a DeviceLoader)')
    p.add_argument('--scaling', choices=['weak', 'strong'], default=os.environ.get('DIC_BENCH_SCALING', 'weak'),
                   help='weak (default): --encounters resident and --batch per step PER GPU.  strong: ONE cohort of --encounters sharded over the '
                        'ranks and a fixed GLOBAL batch of --batch per step (BASELINE configs[2]: 75k encounters, K=8, on 8 GPUs)')
    return p.parse_args()


def launch_ranks(a):
    """``python bench.py --gpus N`` without a launcher around it (the reference's multi-GPU entry is a plain ``python p1...py --num_gpus N``,
    pretrain_trainer.py:21): start the N ranks through torch.distributed.run as CHILD processes -- this process never initialises the GPU
    (no HIP call, no torch.cuda query) -- and relay rank 0's JSON line and the children's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    log(f'--gpus {a.gpus} without a launcher: starting {a.gpus} ranks through torch.distributed.run (port {port})')
    return subprocess.run(cmd, env=env).returncode


def make_args(K, fake_detection=False, dropout=0.0):
    return SimpleNamespace(num_variables=C, num_timestamps=T, ref_points=R, hours_from_admission=H, dropout=dropout,
                           aux_tasks={}, fake_detection=fake_detection, triple_margin=0.0, cluster_number=K,
                           loss='ae_mse_fake_detect_kl' if fake_detection else 'ae_mse_kl',
                           grad_clip=15.0, unsup_aux_tasks={'fake_detection': 1., 'triplet': 1., 'kl': 10.},
                           aux_pos_weights={})


def time_kernel(fn, iters):
    """Average duration (ms) of fn() -- one or more launches on torch's current HIP stream -- by HIP events."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_table(net, x, ob, lengths, K, iters, with_lstm=True, input_path='both'):
    """Per hand-written kernel: algorithmic bytes (SURVEY.md 8d formulas, ragged accounting) / HIP-event time.
    Shapes (C, T, R) are read from the tensors / the net, so BASELINE configs[3] (C=12, T=288) reuses it."""
    from deep_interpolation_clustering_amd import _native as N
    from deep_interpolation_clustering_amd import ops
    L = N.lib()
    B, T = x.shape[0], x.shape[2]
    C = x.shape[1] // 4
    dev = x.device
    grid0 = net.sci.grid()
    R = grid0.numel()
    nsum = float(lengths.sum())                  # sum over (b,c) of observed samples
    grid = net.sci.grid()
    sk, ck, rk = net.sci.kernel.detach(), net.cci.kernel.detach(), net.rbf.kernel.detach()
    st = N.stream_of(x)
    f32 = dict(device=dev, dtype=torch.float32)
    out = torch.empty((B, R, 3 * C), **f32)
    saved = torch.empty((B, 7, C, R), **f32)
    gout = torch.randn((B, R, 3 * C), **f32)
    XW = ops.packed_width(3 * C)               # packed encoder rows: 32 wide for the six vitals, 64 for cfg4's twelve channels (0: not packed)
    xenc_k1 = torch.empty((R, B, max(XW, 8)), device=dev, dtype=torch.bfloat16)
    gx_k1 = torch.randn((R, B, max(XW, 8)), **f32).to(torch.bfloat16)
    gs, gc = torch.empty(C, **f32), torch.empty((C, C), **f32)
    ws1 = torch.empty(max(16, L.dic_sci_cci_bwd_workspace(B, C, R)), dtype=torch.uint8, device=dev)
    v = torch.randn((B, C, R), **f32)
    y, norm, gy = torch.empty((B, C, T), **f32), torch.empty((B, C, T), **f32), torch.randn((B, C, T), **f32)
    gv, gk = torch.empty((B, C, R), **f32), torch.empty(C, **f32)
    ws2 = torch.empty(max(16, L.dic_rbf_bwd_workspace(B, C, T, R)), dtype=torch.uint8, device=dev)
    out2, gl = torch.empty(2, **f32), torch.ones(1, **f32)
    ws3 = torch.empty(max(16, L.dic_masked_sse_workspace(B, C, T)), dtype=torch.uint8, device=dev)
    z, mu = torch.randn((B, D), **f32) * 0.3, torch.randn((K, D), **f32) * 0.3
    q, ts, colsum, p = torch.empty((B, K), **f32), torch.empty((B, K), **f32), torch.empty(K, **f32), torch.empty((B, K), **f32)
    gq, gz, gmu = torch.randn((B, K), **f32), torch.empty((B, D), **f32), torch.empty((K, D), **f32)
    ws4 = torch.empty(max(16, L.dic_dec_fwd_workspace(B, D, K)), dtype=torch.uint8, device=dev)
    ws5 = torch.empty(max(16, L.dic_dec_bwd_workspace(B, D, K)), dtype=torch.uint8, device=dev)
    P = N.ptr
    calls = {
        # (the entry points the bf16 step uses when 3C < 32: the forward writes the encoder LSTM's packed bf16 input rows, the backward
        #  reads the input gradient in that layout; algorithmic bytes as SURVEY.md 8d counts them, for the f32 (B,R,3C) tensors)
        'sci_cci_fwd': ((lambda: L.dic_sci_cci_fwd_packed(P(x), P(lengths), B, C, T, R, P(grid), P(sk), P(ck), None, P(saved), P(xenc_k1), XW, st))
                        if XW else (lambda: L.dic_sci_cci_fwd(P(x), P(lengths), B, C, T, R, P(grid), P(sk), P(ck), P(out), P(saved), st)),
                        8 * nsum + 4 * B * C + 12 * B * C * R),
        'sci_cci_bwd': ((lambda: L.dic_sci_cci_bwd_packed(P(gx_k1), XW, P(saved), P(sk), P(ck), B, C, R, P(gs), P(gc), P(ws1), ws1.numel(), st))
                        if XW else (lambda: L.dic_sci_cci_bwd(P(gout), P(saved), P(sk), P(ck), B, C, R, P(gs), P(gc), P(ws1), ws1.numel(), st)),
                        8 * nsum + 24 * B * C * R),
        'rbf_fwd': (lambda: L.dic_rbf_fwd(P(x), P(lengths), B, C, T, R, P(grid), P(rk), P(v), 0, P(y), P(norm), 1, st),
                    12 * nsum + 4 * B * C * R),
        'rbf_bwd': (lambda: L.dic_rbf_bwd(P(x), P(lengths), B, C, T, R, P(grid), P(rk), P(v), 0, P(y), P(norm), P(gy), P(gv), P(gk),
                                          P(ws2), ws2.numel(), st), 8 * nsum + 8 * B * C * R),
        'masked_sse_fwd': (lambda: L.dic_masked_sse_fwd(P(ob), P(y), None, P(lengths), B, C, T, P(out2), P(ws3), ws3.numel(), st),
                           8 * nsum),
        'masked_sse_bwd': (lambda: L.dic_masked_sse_bwd(P(ob), P(y), None, P(lengths), B, C, T, P(out2), P(gl), P(gy), 1, st),
                           12 * nsum),
        'dec_fwd': (lambda: L.dic_dec_fwd(P(z), P(mu), B, D, K, 1.0, P(q), P(ts), P(colsum), P(ws4), ws4.numel(), st),
                    4.0 * B * (D + 2 * K)),
        'dec_bwd': (lambda: L.dic_dec_bwd(P(z), P(mu), P(q), P(ts), P(gq), B, D, K, 1.0, P(gz), P(gmu), P(ws5), ws5.numel(), st),
                    4.0 * B * (2 * D + 2 * K)),
    }
    # the same k1 / k2 kernels reading the ragged encounter store in place through a shuffled encounter index (what the trainers' DeviceLoader
    # and the timed step above run): packed (t, v) rows, no padded planes
    from deep_interpolation_clustering_amd.ragged import RaggedStore
    stor = RaggedStore(x.detach().cpu().numpy(), C, dev)
    perm = torch.randperm(B, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).to(torch.int32)
    lperm = lengths.index_select(0, perm.to(torch.int64)).contiguous()
    sp = (P(stor.t_pk), P(stor.v_pk))
    out2s = torch.empty(2, **f32)
    ws2s = torch.empty(max(16, L.dic_rbf_fwd_loss_workspace(B, C, T, R)), dtype=torch.uint8, device=dev)
    if XW:
        calls['sci_cci_fwd_store'] = (lambda: L.dic_sci_cci_fwd_store(*sp, None, P(stor.row_off), P(perm), P(lperm), B, C, T, R, P(grid), P(sk), P(ck), None,
                                                                     P(saved), P(xenc_k1), XW, int(stor.times_sorted), st), 8 * nsum + 4 * B * C + 12 * B * C * R)
    calls['rbf_fwd_store'] = (lambda: L.dic_rbf_fwd_store(*sp, P(stor.row_off), P(perm), P(lperm), B, C, T, R, P(grid), P(rk), P(v), 0, 1, P(y), P(norm), 1,
                                                          P(out2s), P(ws2s), ws2s.numel(), st), 12 * nsum + 4 * B * C * R)
    calls['rbf_bwd_store'] = (lambda: L.dic_rbf_bwd_store(*sp, P(stor.row_off), P(perm), P(lperm), B, C, T, R, P(grid), P(rk), P(v), 0, P(y), P(norm), None,
                                                          P(out2s), P(gl), P(gv), P(gk), P(ws2), ws2.numel(), st), 8 * nsum + 8 * B * C * R)
    fwd_first = ['sci_cci_fwd', 'rbf_fwd', 'masked_sse_fwd', 'dec_fwd', 'rbf_fwd_store']
    if with_lstm:
        fwd_first.append('lstm_fwd')
        _lstm_calls(calls, L, P, st, B, R, dev)
    # run the forwards once so the backward inputs (saved, y, norm, out2, ts, LSTM state) hold real values
    # input_path 'store' / 'dense': launch k1 / k2 on one input path only (PMC passes tell kernels apart by name, and both paths share names)
    twin = {'sci_cci_fwd': 'sci_cci_fwd_store', 'rbf_fwd': 'rbf_fwd_store', 'rbf_bwd': 'rbf_bwd_store'}
    drop = set(twin.values()) if input_path == 'dense' else (set(twin.keys()) if input_path == 'store' else set())
    calls = {k: v for k, v in calls.items() if k not in drop}
    for name in fwd_first:
        if name in calls:
            assert calls[name][0]() == 0, name
    table = {}
    for name, (fn, nbytes) in calls.items():
        ms = time_kernel(fn, iters)
        table[name] = {'ms': round(ms, 5), 'algorithmic_bytes': int(nbytes), 'GBps': round(nbytes / ms / 1e6, 1),
                       'frac_hbm_peak': round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4)}
    return table


def _lstm_calls(calls, L, P, st, B, R, dev):
    """persistent bi-LSTM recurrence kernels (decoder shape; the encoder differs only in the GEMM feeding gx)"""
    f32 = dict(device=dev, dtype=torch.float32)
    Hh, bf = 128, torch.bfloat16
    Bp = (B + 63) // 64 * 64
    gxl = (torch.randn((R, B, 2, 4, Hh), **f32) * 0.5).to(bf)
    whh = (torch.randn((2, 4 * Hh, Hh), **f32) * 0.08).to(bf)
    whh_t = whh.transpose(1, 2).contiguous()
    lout, lgates = torch.empty((R, B, 2 * Hh), device=dev, dtype=bf), torch.empty((R, Bp, 2, 4, Hh), device=dev, dtype=bf)
    lcs, lhn, lcn = torch.empty((R, Bp, 2, Hh), device=dev, dtype=bf), torch.empty((2, B, Hh), **f32), torch.empty((2, B, Hh), **f32)
    ldout = (torch.randn((R, B, 2 * Hh), **f32) * 0.1).to(bf)
    ldgx, ldh0, ldc0 = torch.empty((R, B, 2, 4, Hh), device=dev, dtype=bf), torch.empty((2, B, Hh), **f32), torch.empty((2, B, Hh), **f32)
    rows = 2.0 * R * B                     # (step, batch row, direction) units
    calls['lstm_fwd'] = (lambda: L.dic_lstm_fwd(P(gxl), 2 * int(B % 64 == 0), P(whh), None, None, R, B, Hh, P(lout), None, P(lhn), P(lcn), P(lgates), P(lcs), 0, 0, st),
                         rows * (4 * Hh * 2 + Hh * 2 + 4 * Hh * 2 + Hh * 2))      # gx in; h, gates, c (bf16 copy) out
    xenc = (torch.randn((R, B, 32), **f32)).to(bf)
    wih = (torch.randn((2, 4 * Hh, 32), **f32) * 0.1).to(bf)
    calls['lstm_fwd_proj'] = (lambda: L.dic_lstm_fwd_proj(P(xenc), P(wih), P(whh), None, None, R, B, Hh, 32, P(lout), None, P(lhn), P(lcn),
                                                          P(lgates), P(lcs), 0, 0, 1, st),
                              R * B * 32 * 2 + rows * (Hh * 2 + 4 * Hh * 2 + Hh * 2))   # x in; h, gates, c out (encoder: no gx; the decoder rectifies h on load)
    ldb = torch.empty((2, 4 * Hh), **f32)
    ws6 = torch.empty(max(16, L.dic_lstm_bwd_workspace(B)), dtype=torch.uint8, device=dev)
    calls['lstm_bwd'] = (lambda: L.dic_lstm_bwd(P(whh_t), P(lgates), P(lcs), None, P(ldout), None, None, R, B, Hh, P(ldgx), P(ldh0),
                                                P(ldc0), P(ldb), P(ws6), ws6.numel(), 0, 0, st),
                         rows * (4 * Hh * 2 + Hh * 2 + Hh * 2 + 4 * Hh * 2))      # gates, c, dout in; dG out
    # one-pass weight-gradient kernels (csrc/dic_lstmgrad.hip): encoder (32-wide packed input, dX fused) and decoder (256-wide input)
    from deep_interpolation_clustering_amd import _native as N
    if R * B >= 32:
        oext = (torch.randn((R + 2, B, 2 * Hh), **f32) * 0.5).to(bf)
        gr = [torch.zeros(4 * Hh, 18, **f32), torch.zeros(4 * Hh, Hh, **f32), torch.zeros(4 * Hh, **f32), torch.zeros(4 * Hh, **f32)]
        grads_e = [g.clone() for g in gr + gr]
        gr[0] = torch.zeros(4 * Hh, 256, **f32)
        grads_d = [g.clone() for g in gr + gr]
        gpe, gpd = N.ptr_array(grads_e), N.ptr_array(grads_d)
        dxp = torch.empty((2, R * B, 32), device=dev, dtype=bf)
        ws7 = torch.empty(max(16, L.dic_lstm_dw_workspace(R, B)), dtype=torch.uint8, device=dev)
        ws8 = torch.empty(max(16, L.dic_lstm_dw_wide_workspace(R, B)), dtype=torch.uint8, device=dev)
        xdec = torch.randn((R, B, 256), **f32).clamp_min(0).to(bf)
        keep = (oext, grads_e, grads_d, dxp, ws7, ws8, xdec)
        calls['lstm_dw'] = (lambda: L.dic_lstm_dw(P(ldgx), P(oext), P(xenc), P(wih), P(dxp), R, B, Hh, 18, 32, gpe, 0, P(ws7), ws7.numel(), st) + 0 * len(keep),
                            rows * 4 * Hh * 2 + R * B * (2 * Hh * 2 + 32 * 2 + 2 * 32 * 2))   # dG once; h, x in; per-direction dX out
        calls['lstm_dw_wide'] = (lambda: L.dic_lstm_dw_wide(P(ldgx), P(oext), P(xdec), 1, R, B, Hh, 256, gpd, 0, P(ws8), ws8.numel(), st) + 0 * len(keep),
                                 rows * 4 * Hh * 2 + R * B * (2 * Hh * 2 + 256 * 2))           # dG once; h, x once
        # resident-weight projections (csrc/dic_rowproj.hip) and the one-pass CompressFC Linear(256,128) backward (csrc/dic_fcgrad.hip)
        wdec = (torch.randn((8 * Hh, 256), **f32) * 0.06).to(bf)
        bdec = (torch.randn((8 * Hh,), **f32) * 0.1).to(bf)
        w1 = (torch.randn((128, 256), **f32) * 0.06).to(bf)
        b1 = (torch.randn((128,), **f32) * 0.1).to(bf)
        zfc, dzfc = torch.empty((R * B, 128), device=dev, dtype=bf), (torch.randn((R * B, 128), **f32) * 0.1).to(bf)
        sums = torch.empty(257, device=dev, dtype=torch.float64)
        dxfc, dw1 = torch.empty((R * B, 256), device=dev, dtype=bf), torch.zeros((128, 256), **f32)
        ws9 = torch.empty(max(16, L.dic_row_proj_stats_workspace(R * B, 128)), dtype=torch.uint8, device=dev)
        ws10 = torch.empty(max(16, L.dic_fc_bwd_workspace(R * B, 256, 128)), dtype=torch.uint8, device=dev)
        keep2 = (wdec, bdec, w1, b1, zfc, dzfc, sums, dxfc, dw1, ws9, ws10)
        calls['row_proj'] = (lambda: L.dic_row_proj(P(xdec), P(wdec), P(bdec), R * B, 256, 8 * Hh, P(ldgx), B if B % 64 == 0 else 0, 1, st) + 0 * len(keep2),
                             R * B * (256 * 2 + 8 * Hh * 2))                                   # x once; gx out
        # the decoder's forward as the step runs it: the input projection inside the recurrence kernel (no gx): x read once per direction; h, gates, c out
        calls['lstm_fwd_xproj'] = (lambda: L.dic_lstm_fwd_xproj(P(xdec), P(wdec), P(whh), P(bdec), None, None, R, B, Hh, 256, P(lout), None, P(lhn), P(lcn),
                                                                P(lgates), P(lcs), 0, 1, st) + 0 * len(keep2),
                                   2 * R * B * 256 * 2 + rows * (Hh * 2 + 4 * Hh * 2 + Hh * 2))
        calls['row_proj_stats'] = (lambda: L.dic_row_proj_stats(P(xdec), P(w1), P(b1), R * B, 256, 128, P(zfc), P(sums), P(ws9), ws9.numel(), st),
                                   R * B * (256 * 2 + 128 * 2))                                # x once; z out (+ the column sums)
        # the decoder's input gradient dX = dG . W_ih (round 5: 256 x 256 macro-tiles, csrc/dic_dxproj.hip; a library GEMM until round 4): dG once, dX out
        wih_t = (torch.randn((256, 8 * Hh), **f32) * 0.06).to(bf)
        dxd = torch.empty((R * B, 256), device=dev, dtype=bf)
        if R * B >= 256:
            calls['lstm_dx_tile'] = (lambda: L.dic_lstm_dx_tile(P(ldgx), P(wih_t), R * B, 8 * Hh, 256, P(dxd), st) + 0 * len(keep2),
                                     R * B * (8 * Hh * 2 + 256 * 2))
        calls['fc_bwd'] = (lambda: L.dic_fc_bwd(P(dzfc), P(xdec), P(w1), R * B, 256, 128, P(dxfc), P(dw1), P(ws10), ws10.numel(), st),
                           R * B * (128 * 2 + 256 * 2 + 256 * 2))                               # dz, x in; dx out


# ------------------------------------------------------------------------------------------ step trace
def _short(name):
    if name.startswith('Cijk') or name.startswith('Custom_Cijk'):
        mt = name.split('_MT')[1].split('_')[0] if '_MT' in name else '?'
        sk = '_SK' + name.split('_SK')[1].split('_')[0] if '_SK' in name else ''          # stream-K variants are different kernels
        return 'gemm:' + name.split('_')[1 if name.startswith('Cijk') else 2] + '_' + name.split('_')[2 if name.startswith('Cijk') else 3] + '_MT' + mt + sk
    n = name.replace('void ', '')
    if n.startswith('_ZN3dic'):
        import re
        m = re.match(r'_ZN3dic(\d+)', n)
        k = int(m.group(1))
        return 'dic::' + n[len(m.group(0)):len(m.group(0)) + k]
    for cut in ('(', '<'):
        if n.startswith('dic::') and cut in n:
            n = n.split(cut)[0]
    return n[:72]


def _group(name):
    if name.startswith('dic::lstm_dw'):
        return 'lstm_weight_grad'
    if name.startswith('dic::lstm_'):
        return 'lstm_recurrence'
    if name.startswith('dic::'):
        return 'hip_kernels'
    if name.startswith('gemm:') or 'rocblas' in name.lower() or 'Cijk' in name:
        return 'library_gemm'
    if name.startswith('at::native') or 'elementwise' in name or 'reduce_kernel' in name:
        return 'torch_glue'
    return 'other(copies, fills, rng)'


TRACE_STEPS = 3


def step_trace(one_step, first, n=TRACE_STEPS):
    """GPU time of every kernel of n joint steps, from the ROCm tracer behind torch.profiler (the same per-dispatch durations
    rocprofv3 --kernel-trace reports), aggregated per step: {kernel: launches/step, us/launch, ms/step} and per group."""
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for i in range(n):
            one_step(first + i)
        torch.cuda.synchronize()
    wall_us = 1e6 * (time.perf_counter() - t0)
    agg = {}
    for ev in prof.events():
        if ev.device_type != DeviceType.CUDA:
            continue
        dur = float(getattr(ev, 'device_time_total', 0.0) or getattr(ev, 'cuda_time_total', 0.0))
        a = agg.setdefault(_short(ev.name), [0, 0.0])
        a[0] += 1
        a[1] += dur
    traced_us = sum(v[1] for v in agg.values())
    if traced_us < 0.02 * wall_us:          # (launch-bound small batches legitimately sit near 0.2)
        # (seen under rocprofv3: the tracer behind torch.profiler then reports microsecond durations for millisecond kernels)
        raise RuntimeError(f'tracer reports {traced_us / 1e3:.2f} ms of kernels in {wall_us / 1e3:.2f} ms of steps: another profiler is attached?')
    kernels = {k: {'launches_per_step': round(v[0] / n, 2), 'us_per_launch': round(v[1] / v[0], 2), 'ms_per_step': round(v[1] / n / 1e3, 4)}
               for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    groups = {}
    for k, v in kernels.items():
        g = groups.setdefault(_group(k), {'ms_per_step': 0.0, 'launches_per_step': 0.0})
        g['ms_per_step'] = round(g['ms_per_step'] + v['ms_per_step'], 4)
        g['launches_per_step'] = round(g['launches_per_step'] + v['launches_per_step'], 2)
    return kernels, groups


# ------------------------------------------------------------------------------------------ secondary records
FLOP_PER_ENCOUNTER = 3 * (2 * 4 * 128 * ((18 + 128) + (256 + 128)) * 2 * R) + 3 * 2 * R * (256 * 128 + 128 * C)   # bi-LSTMs + CompressFC, fwd+bwd


def record_small_batch(make_stepper, X, OB, LEN, batch=256, steps=100):
    """The reference's own batch size (p1_pretrain_main.py:43): eager launches and one hipGraph replay per step."""
    out = {'per_gpu_batch': batch}
    for mode, graphs in (('eager', False), ('hipgraph', True)):
        st = make_stepper(graphs)
        nb = X.shape[0] // batch

        def one(i):
            lo = (i % nb) * batch
            return st.step(X[lo:lo + batch], OB[lo:lo + batch], None, LEN[lo:lo + batch])
        for i in range(15):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            one(15 + i)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        out[mode] = {'ms_per_step': round(ms, 4), 'encounters_per_s': round(batch / ms * 1e3, 1)}
        if not graphs:          # what the step launches at this batch size: the library-GEMM share must be zero on the bf16 small-batch path
            try:
                kernels, groups = step_trace(one, 0, 2)
                out['launches_per_step'] = round(sum(g['launches_per_step'] for g in groups.values()), 1)
                out['library_gemm_ms'] = groups.get('library_gemm', {}).get('ms_per_step', 0.0)
                out['library_gemm_kernels'] = [k for k in kernels if _group(k) == 'library_gemm']
            except Exception as e:
                out['step_trace_error'] = repr(e)[:200]
        del st
    return out


def record_fake_detection(K, dev, X, OB, LEN, batch, steps=12, warmup=4):
    """Upstream's default unsupervised objective (p3:78 without the private supervised labels: ae_mse + fake detection + 10 kl): sci / cci /
    encoder run a second time on corrupted samples (dataloader.py:182-193) and the detection head is trained.  ms per step, same batch."""
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    args = make_args(K, True, 0.0)
    torch.manual_seed(1234)
    net = Net(args, dev).to(dev)
    net.train()
    st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=torch.bfloat16, use_graphs=False)
    g = torch.Generator(device=dev).manual_seed(5)
    x, ob, ln = X[:batch], OB[:batch], LEN[:batch]
    m = x[:, C:2 * C] > 0
    hit = m & (torch.rand(m.shape, device=dev, generator=g) < 0.5)
    xf = x.clone()
    xf[:, :C] = torch.where(hit, torch.rand(m.shape, device=dev, generator=g) * 5.0 - 2.5, x[:, :C])
    label2 = torch.cat([torch.ones(batch, device=dev), torch.zeros(batch, device=dev)])

    def one():
        perm = torch.randperm(2 * batch, device=dev)
        return st.step(x, ob, None, ln, fake_x=xf, fake_perm_idx=perm, fake_det_label=label2[perm].to(torch.int64))
    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    return {'loss': 'ae_mse + fake_detection + 10*kl', 'per_gpu_batch': batch, 'ms_per_step': round(ms, 3), 'encounters_per_s': round(batch / ms * 1e3, 1)}


F32_MFMA_PEAK_TF, BF16_MFMA_PEAK_TF = 157.3, 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: dense matrix-core peaks


def f32_roofline(kernels, batch, products):
    """Roofline of the dominant kernel of an f32 step (``products`` 'exact' / 'x3') from its step trace.  The 32-row recurrence kernels
    (csrc/dic_lstm32.hip) carry f32 tensors: per (step, row, direction) unit the forward reads gx (4H f32) and writes h, the four gates and c
    (5 120 B); the backward reads gates, c_prev, dL/dout and writes dG (5 120 B); both do 2 x 4H x H flops of recurrent product -- on
    v_mfma_f32_32x32x2_f32 ('exact': priced against the f32 matrix-core peak) or as three bf16 MFMAs ('x3': 3 x the flops against the bf16
    peak).  ``frac`` is the larger of the two fractions and ``bound`` says which one it is."""
    units = 2.0 * R * batch
    rows = []
    for name, v in kernels.items():
        if name.startswith('dic::lstm_rec_'):
            rows.append((v['ms_per_step'], name, v))
    if not rows:
        return None
    _, name, v = max(rows)
    ms = v['us_per_launch'] / 1e3
    nbytes = units * 5120.0
    flop = units * 2.0 * 512 * 128
    gbps = nbytes / ms / 1e6
    if products == 'x3':
        tf, peak_tf = 3.0 * flop / ms / 1e9, BF16_MFMA_PEAK_TF
    else:
        tf, peak_tf = flop / ms / 1e9, F32_MFMA_PEAK_TF
    f_hbm, f_mfma = gbps / HBM_PEAK_GBS, tf / peak_tf
    bound = 'hbm' if f_hbm >= f_mfma else 'mfma'
    traffic = traffic_src = None
    tf_file = os.path.join(ROOT, 'profiles', 'x3_traffic.json')      # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of `--dtype f32x3` (scripts/profile_round.sh, step 8)
    if products == 'x3' and os.path.exists(tf_file):
        tj = json.load(open(tf_file))
        key = name.split('dic::')[-1].split('<')[0].split('(')[0]
        if key in tj and tj.get('_csrc_sha16') == csrc_sha16():      # (only a profile taken on THESE kernel sources)
            traffic = int(tj[key]['hbm_bytes'] * batch / tj.get('_batch', 32768))
            traffic_src = {'from_profile': 'profiles/x3_traffic.json', 'profile_batch': tj.get('_batch', 32768),
                           'note': 'rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of a separate run on these kernel sources, scaled by batch; not measured in this run'}
    return {'kernel': name, 'bound': bound, 'achieved': round(gbps if bound == 'hbm' else tf, 1), 'peak': HBM_PEAK_GBS if bound == 'hbm' else peak_tf,
            'unit': 'GB/s' if bound == 'hbm' else 'TFLOP/s', 'frac': round(max(f_hbm, f_mfma), 4), 'frac_hbm': round(f_hbm, 4), 'frac_mfma': round(f_mfma, 4),
            'ms_per_launch': round(ms, 4), 'launches_per_step': v['launches_per_step'], 'algorithmic_bytes_per_launch': int(nbytes),
            'matrix_core_flop_per_launch': int(flop * (3 if products == 'x3' else 1)), 'traffic': traffic, 'traffic_source': traffic_src,
            'duration_source': 'in-step: per-dispatch durations of this kernel in a trace of the timed steps'}


def record_f32(make_stepper, one_batch, nb, batch, products, steps=20, warmup=4):
    """The f32 step -- every tensor f32, the reference's own arithmetic (clustering_interp.py:14-41, dataloader.py:204) -- at the HEADLINE batch on
    the headline's input path (the ragged store read in place): ``products`` 'exact' = exact-f32 MFMA recurrence + f32 library GEMMs (the 1e-5
    parity configuration of the test suite), 'x3' = every dense product a three-term bf16 split on the matrix cores (csrc/dic_gemm.hip, the
    split recurrence kernels; no library GEMM).  With its own step trace and roofline."""
    st = make_stepper(products)

    def one(i):
        return st.step(*one_batch(i % nb))
    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + i)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    out = {'products': products, 'per_gpu_batch': batch, 'steps': steps, 'ms_per_step': round(ms, 3), 'encounters_per_s': round(batch / ms * 1e3, 1),
           'input': 'ragged encounter store read in place (the headline\'s input path)'}
    try:
        kernels, groups = step_trace(one, warmup + steps, 2)
        out['step_trace'] = {'groups': groups, 'top_kernels': dict(list(kernels.items())[:14])}
        out['library_gemm_ms'] = groups.get('library_gemm', {}).get('ms_per_step', 0.0)
        out['roofline'] = f32_roofline(kernels, batch, products)
        gflop = FLOP_PER_ENCOUNTER * batch / 1e9
        out['whole_step_tflops(useful)'] = round(gflop / ms, 1)
    except Exception as e:
        out['step_trace_error'] = repr(e)[:200]
    return out


def record_loss_deviation(K, dev):
    """First-step losses of the bench's bf16 mode and of the f32 mode against the CPU oracle on the same weights / batch
    (B=256): the bf16 number is the distance of the headline configuration from the 1e-5 parity configuration."""
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    from oracle import dic_oracle as O
    coh = synthetic.make_cohort(256, C=C, T=T, H=H, lam=LAM, G=K, seed=77)
    x_np, ob_np, n = synthetic.stacked_batch(coh)
    x, ob = torch.tensor(x_np), torch.tensor(ob_np)
    args = make_args(K)
    out = {}
    for mode, dt, prec in (('f32', None, 'exact'), ('f32x3', None, 'x3'), ('bf16', torch.bfloat16, None)):
        torch.manual_seed(0)
        ref = O.OracleNet(C, R, H, K, 0.0)
        with torch.no_grad():       # the p3 regime: centroids sit on the latents' clusters (here: phenotype means), KL is O(0.1) and well conditioned
            zs, gl = ref.encode(x)[3], torch.tensor(coh['phenotype'].astype(np.int64))
            ref.cluster_assignment.cluster_centers.copy_(torch.stack([zs[gl == j].mean(0) for j in range(K)]))
        ref.train()
        net = Net(args, dev).to(dev)
        net.load_state_dict(ref.state_dict(), strict=True)
        net.train()
        rterms, _, _ = O.train_step(ref, O.make_optimizer(ref), x, ob, x[:, C:2 * C], 10.0, 15.0)
        st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args, autocast_dtype=dt, precision=prec)
        losses, _, _ = st.step(x.to(dev), ob.to(dev), None, torch.tensor(n, device=dev))
        torch.cuda.synchronize()
        out[mode] = {k: float(abs(float(losses[k].detach()) - rterms[k]) / max(abs(rterms[k]), 1e-30)) for k in ('loss', 'ae_mse', 'kl')}
        out[mode]['oracle_kl'] = rterms['kl']
    return out


def record_cfg4(dev, iters, batch=8192, n_enc=300000, steps=100, warmup=20):
    """BASELINE configs[3] (interp kernel stress): 300 000 encounters, C=12 channels, ~200 observations per channel (T=288), R=24, K=16.
    The kernel table runs on one padded batch; the JOINT STEP runs on the configured cohort: a 300 000-encounter ragged store resident in
    HBM (built on the device, ~6.5 GB; the padded planes would be 16.6 GB), batches of ``batch`` drawn through a per-run randperm over the
    WHOLE cohort as the reference's shuffling loader does (p1_pretrain_main.py:122-131), ``warmup`` + ``steps`` steps."""
    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    C4, T4, LAM4, K4 = 12, 288, 200.0, 16
    coh = synthetic.make_cohort(batch, C=C4, T=T4, H=H, lam=LAM4, G=K4, seed=4)
    x_np, ob_np, len_np = synthetic.stacked_batch(coh)
    x, ob, ln = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(len_np, device=dev)
    a = make_args(K4)
    a.num_variables, a.num_timestamps = C4, T4
    net = Net(a, dev).to(dev)
    table = kernel_table(net, x, ob, ln, K4, iters, with_lstm=False)
    out = {'workload': f'{n_enc} encounters, C={C4}, T={T4}, ~{int(LAM4)} obs/channel, R={R}, K={K4}; batches of {batch}', 'kernels_one_padded_batch': table}
    del x, ob, ln, net, x_np, ob_np, coh
    # the JOINT STEP at this shape (encoder input 3C = 36: 64-wide packed rows through k1 -> fused-projection recurrence -> one-pass dW), bf16 mode,
    # ragged store input; pinned against the reference by tests/test_gpu_traj.py::test_joint_step_wide_shape_K16 (f32) / ..._bf16_tracks_f32
    try:
        from deep_interpolation_clustering_amd.ragged import RaggedBatch
        from deep_interpolation_clustering_amd.step import Stepper
        from deep_interpolation_clustering_amd.utils import pytorch_optimizer
        t0 = time.perf_counter()
        store, _ = synthetic.device_cohort_store(n_enc, C4, T4, H, LAM4, K4, 4, dev)
        torch.cuda.synchronize()
        out['cohort'] = {'encounters': store.N, 'store_GB': round(store.nbytes() / 1e9, 2), 'padded_GB': round(store.N * 4 * C4 * T4 * 4 / 1e9, 1),
                         'mean_obs_per_channel': round(float(store.lengths.float().mean()), 1), 'built_on_device_s': round(time.perf_counter() - t0, 1)}
        perm = torch.randperm(store.N, device=dev, generator=torch.Generator(device=dev).manual_seed(2)).to(torch.int32)
        lens = store.lengths.index_select(0, perm.to(torch.int64)).contiguous()
        nb = store.N // batch

        def run(dtype, precision, n_warm, n_steps):
            torch.manual_seed(1234)
            net = Net(a, dev).to(dev)
            net.train()
            st = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), a, autocast_dtype=dtype, precision=precision)

            def one(i):
                lo = (i % nb) * batch
                return st.step(RaggedBatch(store, perm[lo:lo + batch], lens[lo:lo + batch]), None, None)
            for i in range(n_warm):
                one(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(n_steps):
                one(n_warm + i)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t1) / n_steps
            rec = {'per_gpu_batch': batch, 'dtype': 'bf16' if dtype is not None else 'f32' + (precision or ''), 'steps': n_steps, 'warmup': n_warm,
                   'ms_per_step': round(ms, 3), 'encounters_per_s': round(batch / ms * 1e3, 1),
                   'input': f'ragged store of {store.N} encounters, batches through a randperm of the whole cohort ({nb} batches per epoch)'}
            kernels, groups = step_trace(one, n_warm + n_steps, 2)
            rec['step_trace'] = {'groups': groups, 'top_kernels': dict(list(kernels.items())[:16])}
            rec['library_gemm_ms'] = groups.get('library_gemm', {}).get('ms_per_step', 0.0)
            return rec, kernels
        rec, kernels = run(torch.bfloat16, None, warmup, steps)
        units = 2.0 * R * batch
        dom = max(((v['ms_per_step'], k, v) for k, v in kernels.items() if k.startswith('dic::')), default=None)
        if dom is not None:
            _, k, v = dom
            per_unit = {'dic::lstm_bwd8_kernel': 2560, 'dic::lstm_bwd_kernel': 2560, 'dic::lstm_fwd8_gxn_kernel': 2560, 'dic::lstm_fwd_kernel': 2560,
                        'dic::lstm_fwd8_proj_kernel': 1536 + 128}.get(k.split('<')[0])
            rec['dominant_kernel'] = {'kernel': k, 'ms_per_launch': round(v['us_per_launch'] / 1e3, 4), 'launches_per_step': v['launches_per_step']}
            if per_unit:
                nb_ = units * per_unit
                rec['dominant_kernel'].update({'algorithmic_bytes_per_launch': int(nb_), 'frac_hbm_peak': round(nb_ / (v['us_per_launch'] / 1e3) / 1e6 / HBM_PEAK_GBS, 4)})
        # k1 / k2 inside the step, on batches that do NOT sit in the Infinity Cache from the previous step (algorithmic bytes: SURVEY.md 8d, this batch's mean lengths)
        nsum = float(lens[:batch].sum())
        alg = {'dic::sci_cci_fwd_kernel': 8 * nsum + 4 * batch * C4 + 12 * batch * C4 * R, 'dic::rbf_fwd_row_kernel': 12 * nsum + 4 * batch * C4 * R,
               'dic::rbf_bwd_slot_kernel': 8 * nsum + 8 * batch * C4 * R, 'dic::sci_cci_bwd_kernel': 8 * nsum + 24 * batch * C4 * R}
        rec['interp_kernels_in_step'] = {k.split('dic::')[1]: {'ms': round(v['us_per_launch'] / 1e3, 4), 'algorithmic_bytes': int(alg[k.split('<')[0]]),
                                                               'frac_hbm_peak': round(alg[k.split('<')[0]] / (v['us_per_launch'] / 1e3) / 1e6 / HBM_PEAK_GBS, 4)}
                                         for k, v in kernels.items() if k.split('<')[0] in alg}
        out['step'] = rec
        x3, _ = run(None, 'x3', 5, 20)
        out['step']['f32x3'] = {k: x3[k] for k in ('ms_per_step', 'encounters_per_s', 'library_gemm_ms', 'steps', 'warmup')}
        out['step_f32x3_trace'] = x3['step_trace']
    except Exception as e:
        out.setdefault('step', {})['error'] = repr(e)[:300]
    return out


def record_cfg5(dev, sweep=True):
    """BASELINE configs[4]: k-means on 75 000 x 256 latents -- one Lloyd iteration (all restarts in one launch), whole fits next
    to scikit-learn on this host, and the p2 K = 2..20 sweep (elbow + gap statistic + 3 indices) end to end."""
    import tempfile
    from deep_interpolation_clustering_amd import _native as N
    from deep_interpolation_clustering_amd.kmeans import KMeans
    from deep_interpolation_clustering_amd.synthetic import latent_blobs
    n = 75000
    X, _ = latent_blobs(2024, n, 256, 4, spread=0.35, noise=0.3)
    Xd = torch.tensor(X, device=dev)
    L = N.lib()
    out = {'latents': f'{n} x 256 f32 (77 MB: resident in the 256 MB Infinity Cache)', 'lloyd_iter': {}, 'fit': {}}
    Xc = Xd - Xd.mean(0)
    xn = (Xc * Xc).sum(1)
    for Kk, runs in ((4, 20), (16, 10)):
        cent = Xc[torch.randint(0, n, (runs, Kk), device=dev)].contiguous()
        labels = torch.full((runs, n), -1, dtype=torch.int32, device=dev)
        status = torch.zeros((runs, 8), device=dev)
        status[:, 7] = 1e9
        ws = torch.empty(L.dic_kmeans_workspace(n, 256, Kk, runs), dtype=torch.uint8, device=dev)
        st = N.stream_of(Xc)
        cent0 = cent.clone()

        def reset():
            # every timed launch starts from the same state: fresh random-point centres, no labels, status cleared -- a restart that has
            # converged sets its done flag and later launches skip it, which would bill bytes for work that is not done (round 2's 1.67)
            cent.copy_(cent0)
            labels.fill_(-1)
            status.zero_()
            status[:, 7] = 1e9

        def launch():
            return L.dic_kmeans_lloyd_iter(N.ptr(Xc), N.ptr(xn), n, 256, Kk, runs, N.ptr(cent), N.ptr(labels), N.ptr(status), N.ptr(ws), ws.numel(), st)
        iters, tot, active = 20, 0.0, []
        for it in range(3 + iters):
            reset()
            launch()                                     # iteration 1 from the random-point centres (every point changes label)
            n_act = int((status[:, 0] == 0).sum())       # (host sync) restarts that will do a full iteration in the timed launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch()                                     # iteration 2: the timed one -- all restarts still active (checked below)
            e1.record()
            torch.cuda.synchronize()
            if it >= 3:
                tot += e0.elapsed_time(e1)
                active.append(n_act)
        ms = tot / iters
        n_active = min(active)
        nbytes = n_active * n * (4 * 256 + 4)           # billed for the restarts that were still running when the timed launch began
        out['lloyd_iter'][f'K{Kk}_x{runs}_restarts'] = {'ms': round(ms, 5), 'algorithmic_bytes': nbytes, 'GBps': round(nbytes / ms / 1e6, 1),
                                                       'frac_hbm_peak': round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4), 'active_restarts': n_active,
                                                       'note': 'second Lloyd iteration from fresh random-point centres, state reset before every timed '
                                                               'launch; X (77 MB) is re-read by every restart from the 256 MB Infinity Cache / L2, not '
                                                               'from HBM: a fraction of the HBM peak above ~0.8 here is cache bandwidth'}
    from sklearn.cluster import KMeans as SK
    from threadpoolctl import threadpool_limits
    cores = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get('DIC_CPU_THREADS', 16))))
    for Kk, n_init in ((4, 20), (16, 10)):
        np.random.seed(7529)
        KMeans(n_clusters=Kk, n_init=n_init).fit(Xd)
        torch.cuda.synchronize()
        np.random.seed(7529)
        t0 = time.perf_counter()
        km = KMeans(n_clusters=Kk, n_init=n_init).fit(Xd)
        torch.cuda.synchronize()
        hip_s = time.perf_counter() - t0
        with threadpool_limits(limits=cores):
            np.random.seed(7529)
            t0 = time.perf_counter()
            sk = SK(n_clusters=Kk, n_init=n_init).fit(X)
            sk_s = time.perf_counter() - t0
        out['fit'][f'K{Kk}_n_init{n_init}'] = {'hip_ms': round(1e3 * hip_s, 2), 'sklearn_ms': round(1e3 * sk_s, 1), 'sklearn_threads': cores,
                                               'inertia_rel_diff': float(abs(km.inertia_ - sk.inertia_) / sk.inertia_)}
    if sweep:
        from deep_interpolation_clustering_amd import p2_clustering_optK as p2
        cwd = os.getcwd()
        run = tempfile.mkdtemp(prefix='dic_p2_')
        try:
            os.chdir(run)
            folder = os.path.join(run, 'Results', 'Pretrain', 'out_feat', 'ae_mse')
            os.makedirs(folder)
            for cohort, (m, seed) in {'training': (n, 1), 'validation': (n // 8, 2), 'testing': (n // 8, 3)}.items():
                Xs, _ = latent_blobs(seed, m, 256, 4, centers_seed=99)
                np.save(os.path.join(folder, cohort + '.npy'), {'encounter_id': np.arange(m), 'hidden': Xs, 'ob': np.zeros((m, 1, 1), np.float32),
                                                                'padding_mask': np.ones((m, 1, 1), np.float32)})
            a = p2.get_arguments(['--k_max', '20', '--n_init', '10', '--gap_b', '10'])
            a.restore_metric = ['ae_mse']
            import logging
            logging.disable(logging.INFO)        # (the sweep logs a line per K: stderr stays short for the driver's tail)
            t0 = time.perf_counter()
            try:
                res = p2.main(a)['ae_mse']
            finally:
                logging.disable(logging.NOTSET)
            out['p2_sweep'] = {'seconds': round(time.perf_counter() - t0, 2), 'k_range': '2..20', 'n_init': 10, 'gap_b': 10,
                               'k_by_gap': int(res['gap_sts']['k'][res['gap_sts']['gap'].idxmax()]),
                               'k_by_silhouette': int(res['gap_sts']['k'][res['gap_sts']['Sihouette'].idxmax()])}
        finally:
            os.chdir(cwd)
    return out


def guarded(fn, *a, **kw):
    try:
        return fn(*a, **kw)
    except Exception as e:          # a secondary record must never cost the contract line
        log('secondary record failed:', fn.__name__, repr(e))
        return {'error': repr(e)[:300]}


def cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.lower().startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or 'unknown'


def cpu_quota_cores():
    """CPU time this process may use, in cores: the cgroup quota when there is one (the GPU box shows 256 cores in the affinity mask but grants
    a 16-core share: 256 threads on it take minutes per step), else None."""
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    return max(1, int(round(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    return max(1, int(round(q / int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read()))))
        except (OSError, ValueError, IndexError):
            pass
    return None


def cpu_baseline(K, seconds, B=256, threads=None, max_steps=200, all_cores=False):
    """The CPU oracle (a port of the reference's PyTorch path: oracle/dic_oracle.py) timed on this host.
    B = 256 is the reference's own batch size (p1_pretrain_main.py:43); SURVEY.md 8d also asks for B = 2048 and for 8 threads."""
    from deep_interpolation_clustering_amd import synthetic
    from oracle import dic_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = cpu_quota_cores()
    if all_cores:          # every core this process may actually use: the affinity mask, capped by the cgroup quota (never oversubscribe: see cpu_quota_cores)
        cores = max(1, min(cores, quota or cores, int(os.environ.get('DIC_CPU_THREADS_MAX', 64))))
    else:
        cores = max(1, min(cores, int(os.environ.get('DIC_CPU_THREADS', 16))))    # the GPU box grants ~16 cores per GPU
    if threads:
        cores = max(1, min(cores, threads))
    torch.set_num_threads(cores)
    coh = synthetic.make_cohort(B, C=C, T=T, H=H, lam=LAM, G=K, seed=99)
    x_np, ob_np, _ = synthetic.stacked_batch(coh)
    x, ob = torch.tensor(x_np), torch.tensor(ob_np)
    torch.manual_seed(0)
    net = O.OracleNet(C, R, H, K, 0.0)
    net.train()
    opt = O.make_optimizer(net)
    for _ in range(2):
        O.train_step(net, opt, x, ob, x[:, C:2 * C], 10.0, 15.0)
    n, t0 = 0, time.perf_counter()
    while True:
        O.train_step(net, opt, x, ob, x[:, C:2 * C], 10.0, 15.0)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= max_steps:
            break
    try:
        host_cores = len(os.sched_getaffinity(0))
    except AttributeError:
        host_cores = os.cpu_count() or 1
    return {'value': round(B * n / el, 1), 'unit': 'encounters/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model(),
            'cores_available': host_cores, 'cpu_quota_cores': quota,
            'port_note': 'oracle/dic_oracle.py: the reference path restated with broadcasting ops (fewer (B,C,T,R) temporaries than upstream\'s '
                         'repeat / log / exp sequence): if anything FASTER than the reference\'s own modules on the same cores',
            'sample': f'{n} joint steps of B={B} (C={C}, T={T}, R={R}, K={K}, f32) on torch-CPU, {el:.1f} s',
            'ms_per_step': round(1e3 * el / n, 2)}


CONTRACT_KEYS = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')
CONTRACT_MAX_CHARS = 4096
SECONDARY_FILE = 'bench_secondary.json'


def _finite(o):
    """JSON has no NaN / Infinity: non-finite floats become None (json.dumps(..., allow_nan=False) then never raises)."""
    if isinstance(o, float):
        return o if o == o and abs(o) != float('inf') else None
    if isinstance(o, dict):
        return {str(k): _finite(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_finite(v) for v in o]
    if isinstance(o, (np.floating, np.integer)):
        return _finite(o.item())
    return o


def contract_line(out):
    """The ONE stdout line of the bench contract: exactly CONTRACT_KEYS, compact (< CONTRACT_MAX_CHARS: the driver keeps a bounded tail of
    stdout and parses this line out of it -- round 4's 22 KB line was cut), strict JSON.  Everything else goes to SECONDARY_FILE / stderr."""
    line = json.dumps(_finite({k: out.get(k) for k in CONTRACT_KEYS}), allow_nan=False, separators=(', ', ': '))
    if len(line) >= CONTRACT_MAX_CHARS:        # never lose the line to a verbose sub-record: drop the optional detail, keep the contract fields
        slim = {k: out.get(k) for k in CONTRACT_KEYS}
        slim['roofline'] = {k: v for k, v in (out.get('roofline') or {}).items() if k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic')}
        slim['cpu_baseline'] = {k: v for k, v in (out.get('cpu_baseline') or {}).items() if k in ('value', 'unit', 'cores', 'kind', 'sample')}
        slim['config'] = {k: (v[:200] if isinstance(v, str) else v) for k, v in (out.get('config') or {}).items()}
        line = json.dumps(_finite(slim), allow_nan=False, separators=(', ', ': '))
    assert len(line) < CONTRACT_MAX_CHARS and '\n' not in line, len(line)
    json.loads(line)
    return line


class Secondary:
    """The records beside the contract line (kernel table, step trace, batch sweep, f32 / f32x3 steps, configs[3] / [4] ...): kept in one JSON file
    (rewritten after every record, so a later failure cannot cost the earlier ones) under the repo root and, when there is one, gpurun_out/;
    each record also goes to stderr as one line, cut to a bounded length."""

    def __init__(self, headline):
        self.doc = {'headline': _finite(headline)}
        self.paths = [os.path.join(ROOT, SECONDARY_FILE)]
        if os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
            self.paths.append(os.path.join(ROOT, 'gpurun_out', SECONDARY_FILE))
        self.flush()

    def add(self, key, rec, echo=True):
        self.doc[key] = _finite(rec)
        self.flush()
        if echo:
            txt = json.dumps({key: self.doc[key]}, allow_nan=False)
            log(txt if len(txt) <= 1500 else txt[:1500] + f' ... ({len(txt)} chars: see {SECONDARY_FILE})')

    def flush(self):
        for p in self.paths:
            try:
                with open(p + '.tmp', 'w') as f:
                    json.dump(self.doc, f, allow_nan=False)
                os.replace(p + '.tmp', p)
            except OSError as e:
                log('cannot write', p, repr(e))


def log(*msg):
    print('[bench]', *msg, file=sys.stderr, flush=True)


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources, in name order: profiles/traffic.json / step_traffic.json record the value they were
    measured on (scripts/pmc_traffic.py, step_traffic.py), and a line printed from OTHER sources says traffic: null instead of repeating a
    number that no longer describes the kernels."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'deep_interpolation_clustering_amd', 'csrc')
    for f in sorted(glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(launch_ranks(a))          # (before anything touches the GPU in this process)
    t_start = time.perf_counter()
    from deep_interpolation_clustering_amd import dist, synthetic
    from deep_interpolation_clustering_amd.clustering_interp import Net
    from deep_interpolation_clustering_amd.step import Stepper
    from deep_interpolation_clustering_amd.utils import pytorch_optimizer
    import torch.distributed as td

    rank, world, local = dist.init_from_env()
    sharded = dist.is_sharded()         # world > 1 (or the one-rank RCCL rehearsal of tests/test_gpu_dist.py)
    if sharded and rank == 0:
        log(f'process group: {td.get_backend()}, world {world}')
    if a.gpus != world:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the hot path has no CPU implementation')
    dev = torch.device('cuda', local % torch.cuda.device_count())       # (ranks may share a GPU under DIC_DIST_BACKEND=gloo)
    torch.cuda.set_device(dev)
    K = a.clusters or (8 if world == 8 else 4)
    args = make_args(K, a.fake_detection, a.dropout)

    # ---- cohort shard, resident in HBM before anything is timed
    strong = a.scaling == 'strong'
    if strong:
        # ONE cohort (BASELINE configs[2]: 75k encounters) sharded over the ranks as DeviceLoader shards it -- contiguous row ranges --
        # and a fixed GLOBAL batch: every rank takes global_batch / world rows of its shard per step
        n_total = max(a.encounters, a.batch)
        coh = synthetic.make_cohort(n_total, C=C, T=T, H=H, lam=LAM, G=K, seed=synthetic.SEED)       # same seed on every rank: one cohort
        x_np, ob_np, len_np = synthetic.stacked_batch(coh)
        lo_r, hi_r = dist.shard_bounds(n_total, rank % world, world)
        x_np, ob_np, len_np = x_np[lo_r:hi_r], ob_np[lo_r:hi_r], len_np[lo_r:hi_r]
        a.batch = max(1, a.batch // world)                     # per-rank share of the global batch from here on
        n_enc = hi_r - lo_r
    else:
        n_enc = max(a.encounters, a.batch)
        coh = synthetic.make_cohort(n_enc, C=C, T=T, H=H, lam=LAM, G=K, seed=synthetic.SEED + rank)
        x_np, ob_np, len_np = synthetic.stacked_batch(coh)
    X, OB, LEN = torch.tensor(x_np, device=dev), torch.tensor(ob_np, device=dev), torch.tensor(len_np, device=dev)
    # the cohort as the trainers' DeviceLoader keeps it: a ragged store (observed samples only, packed); a batch = an index range into it
    from deep_interpolation_clustering_amd.ragged import RaggedBatch, RaggedStore
    store = None if a.dense_input else RaggedStore(x_np, C, dev)
    # An EPOCH is the whole cohort (the bench walked 65 536 of its 75 000 resident encounters): ceil(n / batch) batches INCLUDING the
    # short last one -- upstream's DataLoader has no drop_last (p1_pretrain_main.py:122-131) -- behind a fresh permutation per epoch, as the
    # trainers' DeviceLoader draws them (a contiguous slab of the packed store would be the most favourable access pattern for the in-place
    # reads: ~1 KB row groups behind an index -> row_off -> samples chain are what a real epoch sees).  `value` counts the encounters stepped.
    if strong:
        n_enc = n_total // world            # (every rank walks the same number of rows: identical batch sizes, identical collectives)
        X, OB, LEN = X[:n_enc], OB[:n_enc], LEN[:n_enc]
    shuffled = store is not None and not a.fake_detection       # (--fake-detection keeps file order: its corrupted copies XF are a padded tensor in that order)
    perm_gen = torch.Generator(device=dev).manual_seed(7529 + rank)
    nb = (n_enc + a.batch - 1) // a.batch
    nb_full = max(1, n_enc // a.batch)

    def rows_of(i):
        """[lo, hi) of step i's batch inside its epoch's order."""
        lo = (i % nb) * a.batch
        return lo, min(lo + a.batch, n_enc)
    order = {'epoch': -1, 'IDX': torch.arange(n_enc, device=dev, dtype=torch.int32), 'LEN_B': LEN}

    def epoch_order(i):
        e = i // nb
        if shuffled and e != order['epoch']:
            perm = torch.randperm(n_enc, device=dev, generator=perm_gen)
            order['IDX'], order['LEN_B'], order['epoch'] = perm.to(torch.int32), LEN.index_select(0, perm).contiguous(), e      # lengths in batch order
        return order['IDX'], order['LEN_B']
    epoch_order(0)
    del coh, x_np, ob_np
    log(f'rank {rank}: {n_enc} encounters resident after {time.perf_counter() - t_start:.1f}s ({a.scaling} scaling, {a.batch} per step and rank, '
        f'{nb} batches per epoch, the last of {n_enc - (nb - 1) * a.batch})')

    torch.manual_seed(1234)
    net = Net(args, dev).to(dev)
    net.train()
    stepper = Stepper(net, lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4), args,
                      autocast_dtype=torch.bfloat16 if a.dtype == 'bf16' else None, use_graphs=a.graph,
                      precision={'f32': 'exact', 'f32x3': 'x3'}.get(a.dtype))

    XF = label2 = None
    if a.fake_detection:          # corrupted copies as dataloader.py:182-193 makes them (half of each channel's samples -> noise)
        g = torch.Generator(device=dev).manual_seed(5 + rank)
        XF = X.clone()
        m = X[:, C:2 * C] > 0
        hit = m & (torch.rand(m.shape, device=dev, generator=g) < 0.5)
        XF[:, :C] = torch.where(hit, torch.rand(m.shape, device=dev, generator=g) * 5.0 - 2.5, X[:, :C])
        label2 = {}

    def one_step(i):
        lo, hi = rows_of(i)
        IDX, LEN_B = epoch_order(i)
        if store is not None:
            xb, obb = RaggedBatch(store, IDX[lo:hi], LEN_B[lo:hi]), None
        else:
            xb, obb = X[lo:hi], OB[lo:hi]
        g_rows = (hi - lo) * world           # rows of the GLOBAL batch: what a sharded Stepper keys its captured steps on
        if XF is None:
            return stepper.step(xb, obb, None, LEN_B[lo:hi], global_rows=g_rows)
        m = hi - lo
        if m not in label2:
            label2[m] = torch.cat([torch.ones(m, device=dev), torch.zeros(m, device=dev)])
        perm = torch.randperm(2 * m, device=dev)                             # as the trainers draw it (pretrain_trainer.py:156-160)
        return stepper.step(xb, obb, None, LEN_B[lo:hi], fake_x=XF[lo:hi], fake_perm_idx=perm, fake_det_label=label2[m][perm].to(torch.int64),
                            global_rows=g_rows)

    def barrier():
        if sharded:
            td.barrier()
        torch.cuda.synchronize()

    # the steps traced after the timed region (kernel durations INSIDE the step, for the roofline record): one whole epoch from an epoch
    # boundary when an epoch is a handful of batches, else TRACE_STEPS batches
    n_trace = nb if nb <= 8 else TRACE_STEPS
    first_trace = ((a.warmup + a.steps + nb - 1) // nb) * nb
    trace_rows = sum(rows_of(first_trace + i)[1] - rows_of(first_trace + i)[0] for i in range(n_trace))
    for i in range(a.warmup):
        one_step(i)
    barrier()
    t0 = time.perf_counter()
    rows_stepped = 0
    for i in range(a.steps):
        losses, gnorm, _ = one_step(a.warmup + i)
        rows_stepped += rows_of(a.warmup + i)[1] - rows_of(a.warmup + i)[0]
    barrier()
    el = time.perf_counter() - t0
    if sharded:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        el = float(tt)
    final_loss = float(losses['loss'].detach())
    log(f'{a.steps} steps = {rows_stepped} encounters per rank in {el:.3f}s ({1e3 * el / a.steps:.2f} ms/step), loss {final_loss:.5f}')

    if rank == 0:
        ms = 1e3 * el / a.steps
        value = world * rows_stepped / el          # encounters actually stepped (every rank walks the same batch sizes)
        lo = 0
        table = kernel_table(net, X[lo:lo + a.batch], OB[lo:lo + a.batch], LEN[lo:lo + a.batch], K, a.kernel_iters)
        log('kernel table done:', {k: v['ms'] for k, v in table.items()})
        # which kernel dominates the STEP: launches x duration from a trace of the timed step itself
        k2b = ('dic::rbf_bwd_kernel', 'dic::rbf_bwd_wave_kernel', 'dic::rbf_bwd_slot_kernel')
        k2f = ('dic::rbf_fwd_kernel', 'dic::rbf_fwd_row_kernel')
        trace_name = {'sci_cci_fwd': 'dic::sci_cci_fwd_kernel', 'sci_cci_bwd': ('dic::sci_cci_bwd_kernel', 'dic::sci_cci_bwd_lane_kernel'), 'rbf_fwd': k2f,
                      'rbf_bwd': k2b, 'masked_sse_fwd': 'dic::masked_sse_kernel', 'masked_sse_bwd': 'dic::masked_sse_bwd_kernel',
                      'sci_cci_fwd_store': 'dic::sci_cci_fwd_kernel', 'rbf_fwd_store': k2f, 'rbf_bwd_store': k2b,
                      'dec_fwd': 'dic::dec_fwd_kernel', 'dec_bwd': 'dic::dec_bwd_kernel', 'lstm_fwd': ('dic::lstm_fwd8_gxn_kernel', 'dic::lstm_fwd_kernel'),
                      'lstm_fwd_proj': ('dic::lstm_fwd8_proj_kernel', 'dic::lstm_fwd_kernel'), 'lstm_bwd': ('dic::lstm_bwd8_kernel', 'dic::lstm_bwd_kernel'), 'lstm_dw': 'dic::lstm_dw_kernel',
                      'row_proj': 'dic::row_proj_kernel', 'row_proj_stats': 'dic::row_proj_kernel', 'fc_bwd': 'dic::fc_bwd_kernel',
                      'lstm_dw_wide': 'dic::lstm_dw_wide_kernel', 'lstm_fwd_xproj': 'dic::lstm_fwdx8_kernel', 'lstm_dx_tile': 'dic::dx_tile_kernel'}
        kernels = groups = None
        ran = [0]

        def counted_step(i):
            ran[0] += 1
            return one_step(i)
        try:
            kernels, groups = step_trace(counted_step, first_trace, n_trace)    # (sharded: the other ranks run these steps with it, below)
        except Exception as e:
            log('step trace unavailable:', repr(e))
        for i in range(ran[0], n_trace if sharded else 0):                              # stay in lockstep with the other ranks whatever the tracer did
            one_step(first_trace + i)
        # the traced steps are one whole epoch (full batches + the short last one): a kernel's average launch moves trace_rows / n_trace
        # encounters, and its algorithmic bytes per launch are the table's (stated for a full batch; every figure is per encounter) scaled to that
        row_scale = trace_rows / n_trace / a.batch
        per_step = {}
        # k1 / k2 appear twice in the table (padded input, ragged store): the step runs ONE of the two
        twin = {'sci_cci_fwd': 'sci_cci_fwd_store', 'rbf_fwd': 'rbf_fwd_store', 'rbf_bwd': 'rbf_bwd_store'}
        not_in_step = set(twin.keys() if store is not None else twin.values())
        for name, row in table.items():
            launches = 1.0
            if name in not_in_step:
                per_step[name] = 0.0
                continue
            if kernels is not None:
                hits = [v for k, v in kernels.items() if k.startswith(trace_name[name])]        # (str.startswith takes a tuple of prefixes too)
                launches = sum(v['launches_per_step'] for v in hits)
                own = {'lstm_fwd': 'dic::lstm_fwd8_gxn_kernel', 'lstm_fwd_proj': 'dic::lstm_fwd8_proj_kernel'}.get(name)
                if own is not None and any(k.startswith(own) for k in kernels):          # the eight-wave kernels have names of their own
                    launches = sum(v['launches_per_step'] for k, v in kernels.items() if k.startswith(own))
                elif name in ('row_proj', 'row_proj_stats', 'lstm_fwd', 'lstm_fwd_proj'):
                    launches = launches / 2            # one template, two instantiations (decoder / encoder, gx / CompressFC), one launch each
            per_step[name] = launches * row['ms']
        # duration of each table kernel INSIDE the timed step (per-dispatch GPU timestamps of the trace: what rocprofv3 --kernel-trace
        # reports for those launches), where it shares the chip with whatever runs beside it (the decoder's weight-gradient kernel on
        # the side stream); the stand-alone HIP-event figure stays in the table
        in_step_ms = {}
        for name in table:
            if kernels is None:
                break
            if name in not_in_step:
                continue
            own = {'lstm_fwd': 'dic::lstm_fwd8_gxn_kernel', 'lstm_fwd_proj': 'dic::lstm_fwd8_proj_kernel'}.get(name)
            pref = own if (own is not None and any(k.startswith(own) for k in kernels)) else trace_name[name]
            hits = [v for k, v in kernels.items() if k.startswith(pref)]
            if name in ('row_proj', 'row_proj_stats') or (own is None and name in ('lstm_fwd', 'lstm_fwd_proj')):
                continue                               # one kernel name, two shapes: the trace cannot tell their durations apart
            n_l = sum(v['launches_per_step'] for v in hits)
            if n_l > 0:
                in_step_ms[name] = sum(v['ms_per_step'] for v in hits) / n_l
                per_step[name] = n_l * in_step_ms[name]
                table[name]['ms_in_step'] = round(in_step_ms[name], 5)
                table[name]['frac_hbm_peak_in_step'] = round(table[name]['algorithmic_bytes'] * row_scale / in_step_ms[name] / 1e6 / HBM_PEAK_GBS, 4)
        dom = max(per_step, key=per_step.get)
        dom_ms = in_step_ms.get(dom, table[dom]['ms'])
        dom_bytes = table[dom]['algorithmic_bytes'] * (row_scale if dom in in_step_ms else 1.0)      # per AVERAGE launch of the traced epoch
        dom_gbps = dom_bytes / dom_ms / 1e6
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, 'profiles', 'traffic.json')      # PMC-derived HBM bytes per launch (see its _note)
        if os.path.exists(tf):
            tj = json.load(open(tf))
            if dom in tj and tj.get('_csrc_sha16') == csrc_sha16():
                traffic = int(tj[dom]['hbm_bytes'] * (row_scale if dom in in_step_ms else 1.0) * a.batch / tj.get('_batch', a.batch))
                traffic_src = {'from_profile': 'profiles/traffic.json', 'profile_batch': tj.get('_batch'), 'profile_round': tj.get('_round'),
                               'note': 'rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of a separate run ON THESE kernel sources (sha recorded in the file), scaled to the encounters of an average launch; not measured in this run'}
            elif dom in tj:
                log(f"roofline.traffic: null -- profiles/traffic.json was measured on other kernel sources ({tj.get('_csrc_sha16')} != {csrc_sha16()}): re-run scripts/profile_round.sh")
        custom_ms = sum(per_step.values())
        gflop = FLOP_PER_ENCOUNTER * (rows_stepped / a.steps) / 1e9          # (of an average step of the timed loop)
        workload = ((f'{n_enc} of ONE {max(a.encounters, a.batch * world)}-encounter synthetic cohort per GPU' if strong else f'{n_enc} synthetic encounters/GPU') +
                    f', 6 vitals, ~50 irregular samples per channel per 24h (T={T}), R={R}, K={K}, loss ' +
                    ('ae_mse+fake_detect+10*kl' if a.fake_detection else 'ae_mse+10*kl') + ' [BASELINE.json configs[' + ('2' if K == 8 else '1') + ']]' +
                    f'; every epoch walks the WHOLE cohort behind a fresh permutation: {nb} batches of {a.batch} incl. the last of {n_enc - (nb - 1) * a.batch} '
                    f'(no drop_last, as upstream); value = encounters stepped / s; ms_per_step = mean over those steps')
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(dom_gbps, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(dom_gbps / HBM_PEAK_GBS, 4), 'traffic': traffic,
                    'ms_per_launch': round(dom_ms, 5), 'launches_per_step': round(per_step[dom] / dom_ms, 2) if dom_ms else None,
                    'algorithmic_bytes_per_launch': int(dom_bytes), 'encounters_per_launch': round(row_scale * a.batch, 1) if dom in in_step_ms else a.batch,
                    'duration_source': f'in-step per-dispatch GPU timestamps (trace of one epoch = {n_trace} steps of the timed loop)' if dom in in_step_ms else 'stand-alone HIP events',
                    'traffic_source': (f"{traffic_src['from_profile']} (rocprofv3 --pmc, round {traffic_src['profile_round']}, scaled by batch)" if traffic_src else None)}
        roofline_detail = {'frac_standalone': table[dom]['frac_hbm_peak'], 'ms_per_launch_standalone': table[dom]['ms'], 'ms_per_step': round(per_step[dom], 4),
                           'chosen_by': 'launches per step x in-step duration (trace of the timed step)', 'traffic_source': traffic_src,
                           'note': 'frac = algorithmic bytes per launch / the duration this kernel has INSIDE the timed step (per-dispatch GPU timestamps); '
                                   'frac_standalone = HIP events around back-to-back launches on an otherwise idle chip'}
        if a.dtype != 'bf16' and kernels is not None:       # the f32 modes run the 32-row recurrence kernels: their own roofline
            rl = f32_roofline(kernels, row_scale * a.batch, 'x3' if a.dtype == 'f32x3' else 'exact')      # (encounters of an average launch of the traced epoch)
            if rl:
                roofline_detail = {'full': rl}
                roofline = {k: rl[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'ms_per_launch', 'launches_per_step',
                                               'algorithmic_bytes_per_launch', 'frac_hbm', 'frac_mfma')}
        if a.dtype != 'bf16' and kernels is None:
            # the kernel table above times the bf16 kernels: without an in-step trace (another profiler attached) an f32 run has no duration to put a roofline on
            roofline, roofline_detail = None, {'note': 'no in-step trace of the f32 step: see the rocprofv3 kernel statistics of this run'}
        out = {
            'metric': 'encounters/sec per joint interp+DEC step', 'value': round(value, 1), 'unit': 'encounters/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms, 3),
            'higher_is_better': True, 'scaling': a.scaling, 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': workload, 'per_gpu_batch': a.batch, 'global_batch': a.batch * world, 'batches_per_epoch': nb,
                       'encounters_stepped': world * rows_stepped, 'parallelism': f'dp{world}' if world > 1 else 'single',
                       'index_order': 'shuffled' if shuffled else 'file order',
                       'input': 'padded (B,4C,T) batches' if store is None else f'ragged encounter store read in place ({store.nbytes() / 1e6:.0f} MB resident)'},
            'roofline': roofline,
            'cpu_baseline': None,
        }
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(K, a.cpu_seconds)
        # ---- the contract line: the ONLY thing this program writes to stdout, written before any secondary record runs
        print(contract_line(out), flush=True)

        sec = Secondary(out)
        sec.add('dtype_note', {'bf16': 'HIP kernels compute in f32; bf16 = torch.autocast for the bi-LSTMs / FC heads only '
                                       '(outside the 1e-5 parity configuration: see loss_rel_dev_vs_oracle; the f32 / f32x3 records carry the parity-grade rates)',
                               'f32': 'every tensor and product f32: exact-f32 MFMA recurrence, f32 library GEMMs',
                               'f32x3': 'every tensor f32; every dense product a three-term bf16 split (hi.hi + lo.hi + hi.lo) on the bf16 matrix cores '
                                        'with f32 accumulation: no library GEMM; losses within 1e-5 of the reference (loss_rel_dev_vs_oracle)'}[a.dtype], echo=False)
        sec.add('config_detail', {'padded_array_MB': round(n_enc * 4 * C * T * 4 / 1e6),
                                  'index_order': 'shuffled (a per-run randperm of the cohort, as the trainers\' DeviceLoader draws batches)' if shuffled
                                                 else 'file order (contiguous rows)'}, echo=False)
        sec.add('roofline_detail', roofline_detail, echo=False)
        sec.add('kernels', table, echo=False)
        whole = {'gflop_per_step_dense(lstm+fc, fwd+bwd)': round(gflop, 1), 'tflops': round(gflop / ms, 1),
                 'frac_bf16_mfma_peak(2500 TF)': round(gflop / ms / 2500.0, 4), 'hip_kernels_ms(table x launches)': round(custom_ms, 3), 'final_loss': final_loss}
        if groups is not None:
            sec.add('step_trace', {'groups': groups, 'top_kernels': dict(list(kernels.items())[:24]),
                                   'kernel_ms_per_step': round(sum(g['ms_per_step'] for g in groups.values()), 3),
                                   'launches_per_step': round(sum(g['launches_per_step'] for g in groups.values()), 1)}, echo=False)
            gg = groups.get('library_gemm')
            if gg:
                whole['library_gemm_ms'] = gg['ms_per_step']
        st_file = os.path.join(ROOT, 'profiles', 'step_traffic.json')
        if os.path.exists(st_file):
            sj = json.load(open(st_file))
            if 'total_bytes' in sj and sj.get('_csrc_sha16') == csrc_sha16():
                gb = sj['total_bytes'] * (rows_stepped / a.steps) / sj.get('_batch', 32768) / 1e9
                whole['hbm_traffic'] = {'GB_per_step': round(gb, 2), 'TBps_over_step': round(gb / ms, 3), 'from_profile': 'profiles/step_traffic.json',
                                        'profile_round': sj.get('_round')}
        sec.add('whole_step', whole)
        if world == 1 and not a.no_cpu_baseline and not a.no_secondary:
            # the other points of SURVEY.md 8d's CPU comparison, a few seconds each (the last: every core the process may use, capped by the cgroup quota and at 64)
            sec.add('cpu_baseline_more', [guarded(cpu_baseline, K, 6.0, 256, 8, 60), guarded(cpu_baseline, K, 8.0, 2048, None, 12),
                                          guarded(cpu_baseline, K, 8.0, 2048, None, 6, True)])
        if world == 1 and not a.no_secondary:
            del stepper
            torch.cuda.empty_cache()
            opt_f = lambda m: pytorch_optimizer(m, 'Adam', 3e-3, 4e-4)      # noqa: E731

            def fresh(dtype, graphs, precision=None):
                torch.manual_seed(1234)
                n2 = Net(args, dev).to(dev)
                n2.train()
                return Stepper(n2, opt_f, args, autocast_dtype=dtype, use_graphs=graphs, precision=precision)
            sec.add('batch256', guarded(record_small_batch, lambda g: fresh(torch.bfloat16, g), X, OB, LEN))
            # SURVEY.md 8d's throughput sweep (B = 256, 2 048, 16 384 per GPU; 256 is the record above, the headline is 32 768)
            sec.add('batch_sweep', {str(bs): guarded(record_small_batch, lambda g: fresh(torch.bfloat16, g), X, OB, LEN, bs, st_)
                                    for bs, st_ in ((2048, 60), (4096, 40), (16384, 20))})
            # the f32 step at the headline batch on the headline's input path: the reference's own arithmetic ('exact') and the same tensors with
            # every dense product as a three-term bf16 split on the matrix cores ('x3': the parity-grade throughput record)
            if store is not None:
                def f32_batch(i):
                    lo_ = i * a.batch
                    return RaggedBatch(store, order['IDX'][lo_:lo_ + a.batch], order['LEN_B'][lo_:lo_ + a.batch]), None, None
            else:
                def f32_batch(i):
                    lo_ = i * a.batch
                    return X[lo_:lo_ + a.batch], OB[lo_:lo_ + a.batch], None, LEN[lo_:lo_ + a.batch]
            for key, products in (('f32x3', 'x3'), ('f32', 'exact')):
                rec = guarded(record_f32, lambda pr: fresh(None, False, pr), f32_batch, nb_full, a.batch, products)          # (full batches: the x3 / f32 records keep round 5's definition)
                torch.cuda.empty_cache()
                sec.add(key, rec, echo=False)
                log(key, 'done', json.dumps(_finite({k: v for k, v in rec.items() if k in ('ms_per_step', 'encounters_per_s', 'library_gemm_ms', 'error')})))
            sec.add('fake_detection_objective', guarded(record_fake_detection, K, dev, X, OB, LEN, a.batch))
            sec.add('loss_rel_dev_vs_oracle', guarded(record_loss_deviation, K, dev))
            del X, OB
            torch.cuda.empty_cache()
            rec = guarded(record_cfg4, dev, a.kernel_iters)
            sec.add('cfg4', rec, echo=False)
            log('cfg4 done', json.dumps(_finite({k: v for k, v in rec.get('step', rec).items() if k in ('cohort', 'per_gpu_batch', 'ms_per_step', 'encounters_per_s',
                                                                                                          'library_gemm_ms', 'dominant_kernel', 'f32x3', 'error')})))
            sec.add('cfg5', guarded(record_cfg5, dev, not a.no_sweep))
        log('secondary records:', ', '.join(sec.paths))
    elif world > 1:
        # rank 0 traces n_trace more optimisation steps for the roofline record: a sharded step is full of collectives (loss
        # sums, BatchNorm moments, the gradient bucket), so every rank has to run them with it or rank 0 waits forever
        for i in range(n_trace):
            one_step(first_trace + i)
        torch.cuda.synchronize()
    if sharded:
        td.barrier()
        td.destroy_process_group()


if __name__ == '__main__':
    main()
