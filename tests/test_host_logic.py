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
