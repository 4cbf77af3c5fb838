"""Host-side logic that needs no GPU."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_gap_stream_walker_deals_problems_and_replicates_the_stream():
    """p2's sharded gap statistic: every rank walks NumPy's global stream through ALL draws of upstream's loop (p2:353-410: draw(ref) ->
    that fit's k-means++ seeds -> next draw ...) but is handed only its own problems -- with the uniform draws and the seeding state the
    serial loop would have used, and the stream left where the serial loop leaves it."""
    from deep_interpolation_clustering_amd import p2_clustering_optK as p2
    from deep_interpolation_clustering_amd.kmeans import seed_draw_count
    shape, ks, n_ref, n_init = (50, 8), [2, 3, 4], 3, 2
    # the serial loop, spelled with plain NumPy calls
    np.random.seed(11)
    serial = []
    for ki, k in enumerate(ks):
        for i in range(n_ref + 1):
            u = np.random.random_sample(shape) if i < n_ref else None
            st = np.random.get_state()
            rs = np.random.RandomState()
            rs.set_state(st)
            first = rs.random_sample(3)                                   # what a fit seeded from here would see first
            np.random.random_sample(seed_draw_count(k, n_init))
            serial.append((ki, i, u, first))
    end = np.random.random()
    got = {}
    for rank in range(2):
        np.random.seed(11)
        w = p2._StreamWalker(shape, ks, n_ref, n_init, rank, 2)
        for ki, i, buf, st in w:
            got[(ki, i)] = (rank, None if buf is None else buf.copy(), p2._random_state_at(st).random_sample(3))
            if buf is not None:
                w.release(buf)
        w.join()
        assert np.random.random() == end
    assert len(got) == len(serial)
    for j, (ki, i, u, first) in enumerate(serial):
        rank, buf, f = got[(ki, i)]
        assert rank == j % 2
        assert (u is None and buf is None) or np.array_equal(u, buf)
        assert np.array_equal(first, f)


def test_bench_contract_line_is_compact_strict_json():
    """bench.py's ONE stdout line: exactly the contract keys, < 4 KB, strict JSON (no NaN / Infinity) whatever the sub-records hold -- round 4's
    22 KB line was cut by the driver's bounded stdout tail and never parsed."""
    import json

    import bench
    roof = {'bound': 'hbm', 'kernel': 'lstm_bwd', 'achieved': 5378.1, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.6723, 'traffic': 4098000000,
            'ms_per_launch': 0.74885, 'launches_per_step': 2.0, 'algorithmic_bytes_per_launch': 4026531840,
            'duration_source': 'in-step per-dispatch GPU timestamps (trace of the timed step)', 'traffic_source': 'profiles/traffic.json (...)'}
    cpu = {'value': 2011.3, 'unit': 'encounters/s', 'cores': 16, 'kind': 'port', 'cpu_model': 'AMD EPYC 9575F 64-Core Processor', 'cores_available': 256,
           'cpu_quota_cores': 16, 'port_note': 'x' * 300, 'sample': '118 joint steps of B=256 (C=6, T=96, R=24, K=4, f32) on torch-CPU, 15.0 s', 'ms_per_step': 127.3}
    out = {'metric': 'encounters/sec per joint interp+DEC step', 'value': 5710000.0, 'unit': 'encounters/s', 'n_gpus': 1, 'steps': 100, 'warmup': 20,
           'ms_per_step': 5.742, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
           'config': {'workload': '75000 synthetic encounters/GPU, 6 vitals, ...', 'per_gpu_batch': 32768, 'global_batch': 32768, 'parallelism': 'single',
                      'index_order': 'shuffled', 'input': 'ragged encounter store read in place (208 MB resident)'},
           'roofline': roof, 'cpu_baseline': cpu, 'kernels': {'k%d' % i: {'ms': 0.1} for i in range(400)}, 'final_loss': float('nan')}
    line = bench.contract_line(out)
    assert len(line) < 4096 and '\n' not in line
    d = json.loads(line)
    assert tuple(d) == bench.CONTRACT_KEYS
    assert d['roofline']['frac'] == 0.6723 and d['cpu_baseline']['cores'] == 16 and d['config']['per_gpu_batch'] == 32768
    # non-finite numbers never reach the line; an over-long sub-record is cut down to the contract fields instead of costing the line
    out['roofline'] = dict(roof, frac=float('inf'), note='y' * 5000)
    out['ms_per_step'] = float('nan')
    line = bench.contract_line(out)
    d = json.loads(line)
    assert len(line) < 4096 and d['roofline']['frac'] is None and d['ms_per_step'] is None and 'note' not in d['roofline']
    assert set(d['roofline']) == {'bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic'}


def test_bench_secondary_records_go_to_a_side_file(tmp_path, monkeypatch, capsys):
    import json

    import bench
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    sec = bench.Secondary({'value': 1.0})
    sec.add('cfg4', {'ms': float('nan'), 'big': list(range(2000))})
    doc = json.load(open(tmp_path / bench.SECONDARY_FILE))
    assert doc['headline'] == {'value': 1.0} and doc['cfg4']['ms'] is None and len(doc['cfg4']['big']) == 2000
    cap = capsys.readouterr()
    assert cap.out == ''                                     # stdout belongs to the contract line alone
    assert all(len(ln) < 1700 for ln in cap.err.splitlines())


def test_ragged_store_built_on_the_device_and_joined_equals_the_host_built_store():
    """RaggedStore.from_device / concat (cohorts too large to pad on the host) against the constructor on the same planes."""
    import torch

    from deep_interpolation_clustering_amd import synthetic
    from deep_interpolation_clustering_amd.ragged import RaggedStore
    C = 3
    coh = synthetic.make_cohort(37, C=C, T=20, H=24.0, lam=7.0, G=2, seed=5)
    x_np, _, _ = synthetic.stacked_batch(coh)
    ref = RaggedStore(x_np, C, 'cpu')
    x = torch.tensor(x_np)
    whole = RaggedStore.from_device(x, C)
    joined = RaggedStore.concat([RaggedStore.from_device(x[:10].contiguous(), C), RaggedStore.from_device(x[10:11].contiguous(), C),
                                 RaggedStore.from_device(x[11:].contiguous(), C)])
    for s in (whole, joined):
        assert (s.N, s.C, s.T, s.times_sorted) == (ref.N, ref.C, ref.T, ref.times_sorted)
        for name in ('t_pk', 'v_pk', 'hold_pk', 'row_off', 'lengths', 'pad_value'):
            a, b = getattr(s, name), getattr(ref, name)
            assert a.dtype == b.dtype and torch.equal(a, b), name
        idx = torch.tensor([36, 0, 10, 11, 5])
        assert torch.equal(s.dense_rows(idx), ref.dense_rows(idx))
    #  a chunk whose rows are all full-length in a channel has no padded slot there: it says nothing about that channel's constant
    # and must not veto the join; a mask that is not a prefix is refused instead of being packed wrongly
    xf = x.clone()
    xf[:, C:2 * C, :] = 0
    xf[:, C:2 * C, :5] = 1                        # five observed slots everywhere ...
    xf[:4, C, :] = 1                              # ... except that channel 0 of the first four rows is full-length
    pad_const = torch.tensor([-2.5, -2.0, -1.5])
    xf[:, 0:C] = torch.where(xf[:, C:2 * C] != 0, xf[:, 0:C], pad_const[None, :, None].expand(-1, -1, xf.shape[-1]))
    xf[:, 2 * C:3 * C] = xf[:, 2 * C:3 * C] * xf[:, C:2 * C]
    xf[:, 3 * C:4 * C] = xf[:, 3 * C:4 * C] * xf[:, C:2 * C]
    head, rest = RaggedStore.from_device(xf[:4].contiguous(), C), RaggedStore.from_device(xf[4:].contiguous(), C)
    assert not bool(head.has_pad[0]) and bool(rest.has_pad.all())
    both = RaggedStore.concat([head, rest])
    assert torch.equal(both.pad_value, pad_const) and torch.equal(both.dense_rows(torch.arange(37)), RaggedStore(xf.numpy(), C, 'cpu').dense_rows(torch.arange(37)))
    import pytest
    bad = xf.clone()
    bad[0, C + 1, 2] = 0                          # a hole inside the prefix
    with pytest.raises(ValueError, match='not prefix masks'):
        RaggedStore.from_device(bad, C)
    st, ph = synthetic.device_cohort_store(50, C, 20, 24.0, 7.0, 2, 3, 'cpu', chunk=16)
    assert st.N == 50 and ph.shape == (50,) and st.times_sorted and int(st.lengths.min()) >= 1
    assert float(st.v_pk.abs().max()) <= 2.5 and int(st.row_off[-1]) == int(st.lengths.sum())


def test_num_gpus_decides_between_launching_ranks_and_being_one(monkeypatch):
    """dist.ranks_for_num_gpus: upstream's ``--num_gpus N`` (p1_pretrain_main.py:27,118) means N processes here -- a plain start launches them (children
    through torch.distributed.run; the parent relays the exit code), a rank goes on, a contradicting launcher environment is an error."""
    import pytest

    from deep_interpolation_clustering_amd import dist
    calls = []
    monkeypatch.setattr(dist, 'launch_ranks', lambda n, module, argv: calls.append((n, module, list(argv))) or 7)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    assert dist.ranks_for_num_gpus(1, 'pkg.p1', ['--mode', 'train']) is None and dist.ranks_for_num_gpus(0, 'pkg.p1', []) is None and not calls
    assert dist.ranks_for_num_gpus(4, 'pkg.p1', ['--num_gpus', '4', '--mode', 'train']) == 7
    assert calls == [(4, 'pkg.p1', ['--num_gpus', '4', '--mode', 'train'])]
    monkeypatch.setenv('WORLD_SIZE', '4')
    assert dist.ranks_for_num_gpus(4, 'pkg.p1', []) is None                     # a rank of the launched job
    assert dist.ranks_for_num_gpus(1, 'pkg.p1', []) is None                     # under a launcher with the default --num_gpus: the launcher's world
    with pytest.raises(SystemExit, match='--num_gpus 2 but WORLD_SIZE=4'):
        dist.ranks_for_num_gpus(2, 'pkg.p1', [])
    assert len(calls) == 1


def test_launch_ranks_command_line(monkeypatch):
    import subprocess
    import sys

    from deep_interpolation_clustering_amd import dist
    seen = {}

    def fake_run(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return subprocess.CompletedProcess(cmd, 3)
    monkeypatch.setattr(subprocess, 'run', fake_run)
    assert dist.launch_ranks(2, 'pkg.p3', ['--num_gpus', '2']) == 3
    cmd = seen['cmd']
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run'] and '--nproc-per-node=2' in cmd and cmd[-4:] == ['-m', 'pkg.p3', '--num_gpus', '2']
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_environment_switch_table():
    """switches.py: the ONE table of DIC_* variables.  Unknown names and values outside the allowed sets raise (a typo used to mean the default
    silently); every variable the sources read is in the table; INTEGRATION.md section 4 is the generated table."""
    import glob
    import re

    import pytest

    from deep_interpolation_clustering_amd import switches
    switches.validate({'PATH': '/bin', 'DIC_ROW_PROJ': '0', 'DIC_SMALL_BATCH': '2048', 'DIC_RBF_BWD_SLOT': '2', 'DIC_LIB_PATH': '/tmp/x.so'})
    with pytest.raises(RuntimeError, match='DIC_ROW_PROJS: unknown switch .did you mean DIC_ROW_PROJ'):
        switches.validate({'DIC_ROW_PROJS': '0'})
    with pytest.raises(RuntimeError, match="DIC_ROW_PROJ='off': allowed values are 0, 1"):
        switches.validate({'DIC_ROW_PROJ': 'off'})
    with pytest.raises(RuntimeError, match='DIC_SMALL_BATCH=.4k.: an integer'):
        switches.validate({'DIC_SMALL_BATCH': '4k'})
    with pytest.raises(RuntimeError, match='DIC_F32_PRODUCTS'):
        switches.validate({'DIC_F32_PRODUCTS': 'x6'})
    assert switches.get('DIC_RBF_BWD_SLOT', {}) == '1' and switches.get('DIC_RBF_BWD_SLOT', {'DIC_RBF_BWD_SLOT': '0'}) == '0'
    #  an empty value is the switch unset; a DIC_* name that is nobody's typo -- a build-macro name someone exported, a site's own -- warns
    switches.validate({'DIC_LIB_PATH': '', 'DIC_SMALL_BATCH': '', 'DIC_ROW_PROJ': '', 'DIC_REFERENCE': '/somewhere'})
    assert switches.get('DIC_SMALL_BATCH', {'DIC_SMALL_BATCH': ''}) == '4096'
    with pytest.warns(UserWarning, match='DIC_NO_NT, DIC_SITE_QUEUE are not switches'):
        switches.validate({'DIC_NO_NT': '1', 'DIC_SITE_QUEUE': 'a'})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, 'deep_interpolation_clustering_amd')
    read = set()
    files = (glob.glob(os.path.join(pkg, '*.py')) + glob.glob(os.path.join(pkg, 'csrc', '*.h*')) + [os.path.join(root, 'bench.py')]
             + glob.glob(os.path.join(root, 'scripts', '*')) + glob.glob(os.path.join(root, 'oracle', '*.py')))
    for f in files:
        for ln in open(f, errors='replace'):
            if 'getenv' in ln or 'os.environ' in ln:
                read.update(re.findall(r'DIC_[A-Z0-9_]+', ln))
    assert read and read <= set(switches.SWITCHES), sorted(read - set(switches.SWITCHES))
    doc = open(os.path.join(root, 'INTEGRATION.md')).read()
    block = doc[doc.index(switches.BEGIN) + len(switches.BEGIN):doc.index(switches.END)].strip()
    assert block == switches.markdown_table(), 'INTEGRATION.md section 4 is stale: python -m deep_interpolation_clustering_amd.switches --write'


def test_distance_pass_tile_lists_cover_every_block_pair_once():
    """The host side of dic_cluster_intra_totals / dic_cluster_pair_rowsums (cluster_stats._tile_list / _row_tile_list): the totals list holds every pair of
    256-row blocks I <= J of one cluster exactly once, sorted by cluster; the row-sum list every (row block, cluster block) pair, sorted by (row block, cluster,
    block), its slots are the maximal runs of one (row block, cluster) inside a workgroup's contiguous range and group_start indexes them per (row block, cluster)
    -- with an empty cluster, a cluster smaller than a block and more tiles than workgroups."""
    from deep_interpolation_clustering_amd import cluster_stats as cs
    counts = np.array([700, 0, 5, 1300, 256])
    n, K = int(counts.sum()), len(counts)
    seg = np.concatenate([[0], np.cumsum(counts)])
    t = cs._tile_list(counts)
    want = [(seg[c] + 256 * i, seg[c] + 256 * j, seg[c + 1], c) for c in range(K) for i in range(-(-counts[c] // 256)) for j in range(i, -(-counts[c] // 256))]
    assert sorted(map(tuple, t.tolist())) == sorted(want) and (np.diff(t[:, 3]) >= 0).all() and t.dtype == np.int32
    assert cs._tile_list(np.array([0, 0])).shape == (0, 4)

    for workgroups in (256, 7):                       # 9 x 10 = 90 tiles: fewer / more tiles than workgroups
        cs_wg, cs.ROWSUM_WORKGROUPS = cs.ROWSUM_WORKGROUPS, workgroups
        try:
            tiles, group_start, n_slots = cs._row_tile_list(counts, n)
        finally:
            cs.ROWSUM_WORKGROUPS = cs_wg
        n_i = -(-n // 256)
        jb = [(seg[c] + 256 * j, seg[c + 1], c) for c in range(K) for j in range(-(-counts[c] // 256))]
        assert [tuple(r[:3]) for r in tiles.tolist()] == [(256 * i, j0, je) for i in range(n_i) for j0, je, _ in jb]
        group = np.repeat(np.arange(n_i) * K, len(jb)) + np.tile([c for _, _, c in jb], n_i)
        slot = tiles[:, 3]
        assert slot[0] == 0 and set(np.diff(slot)) <= {0, 1} and n_slots == slot[-1] + 1
        nwg = min(len(tiles), workgroups)
        per = -(-len(tiles) // nwg)
        wg = np.arange(len(tiles)) // per
        for s_ in range(n_slots):                     # one (row block, cluster) and one workgroup per slot; neighbours differ in one of the two
            m = slot == s_
            assert len(set(group[m])) == 1 and len(set(wg[m])) == 1
        same = (group[1:] == group[:-1]) & (wg[1:] == wg[:-1])
        assert ((np.diff(slot) == 0) == same).all()
        assert group_start.shape == (n_i * K + 1,) and group_start[0] == 0 and group_start[-1] == n_slots
        for g in range(n_i * K):
            assert sorted(set(slot[group == g])) == list(range(group_start[g], group_start[g + 1]))
