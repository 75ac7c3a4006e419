"""world_size-2 gloo test of the all-reduce hook used by the sharded solve (runs on CPU)."""
import os
import socket
import tempfile

import numpy as np


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_gloo_allreduce_hook_world_size_2():
    import torch.multiprocessing as mp
    import dist_workers
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(dist_workers.hook_on_host_memory, args=(2, _free_port(), d), nprocs=2, join=True)
        r0, r1 = np.load(os.path.join(d, "host_0.npz")), np.load(os.path.join(d, "host_1.npz"))
        expect = np.arange(1000, dtype=np.float64) * 3
        assert np.array_equal(r0["a"], expect) and np.array_equal(r1["a"], expect)
        assert np.array_equal(r0["b"], [1.0, 1.0]) and np.array_equal(r1["b"], [1.0, 1.0])   # slot-gather idiom
        assert int(r0["calls"]) == 2
