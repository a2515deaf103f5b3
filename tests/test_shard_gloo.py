"""N > 1 host path on CPU: world_size-2 gloo job (two processes), sharded windows == unsharded."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import REPO
from incorporating_different_sources_amd import shard


def test_partition_is_contiguous_and_balanced():
    for W in (0, 1, 7, 8, 9, 10000, 200000):
        for world in (1, 2, 3, 4, 8):
            r = shard.partition(W, world)
            assert len(r) == world and r[0][0] == 0 and r[-1][1] == W
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            sizes = [h - l for l, h in r]
            assert max(sizes) - min(sizes) <= 1


def test_slice_window_inputs_rebases_offsets():
    from incorporating_different_sources_amd import synthetic
    inp = synthetic.make_kernel_inputs(5, 20, 11, seed=1)
    s = shard.slice_window_inputs(inp, 4, 9, inp["n_r"], inp["m"])
    assert s["W"] == 5 and s["start"][0] == 0 and s["hf_start"][0] == 0
    for i, w in enumerate(range(4, 9)):
        a = inp["panel"][inp["start"][w]: inp["start"][w] + inp["n_r"]]
        b = s["panel"][s["start"][i]: s["start"][i] + inp["n_r"]]
        assert np.array_equal(a, b)
        a = inp["hf_panel"][inp["hf_start"][w]: inp["hf_start"][w] + inp["m"]]
        b = s["hf_panel"][s["hf_start"][i]: s["hf_start"][i] + inp["m"]]
        assert np.array_equal(a, b)
    assert s["panel"].shape[0] == 4 + inp["n_r"] + 0 + 1 - 1 + 0 or s["panel"].shape[0] == (8 - 4) + inp["n_r"]


def test_world_size_2_gloo_sharded_equals_unsharded(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "result.json"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "_shard_worker.py"), str(out)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()[-2000:]
    res = json.load(open(out))
    assert res["world"] == 2 and res["shape"] == [37, 12] and res["uid_ok"] and res["tmax"] == 2.0
    assert res["max_abs_diff"] == 0.0      # windows are independent: sharding changes nothing, bit for bit
