"""Sharding parity on one GPU box: two ranks (gloo, both on GPU 0: the rehearsal mode of bench.py) run
the sequence workload, each on its own sequence, and the gathered trajectories must be byte-equal to
what single-rank runs of the same two sequences give."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_two_rank_gather_equals_single_rank_runs(tmp_path):
    nf = 8
    common = ["bench.py", "--workload", "sequence", "--frames", str(nf)]
    two = tmp_path / "two.npy"
    out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", "29641"] + common + ["--gpus", "2", "--dump-trajectories", str(two)],
               {"KTLVO_BENCH_REHEARSAL": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert out["n_gpus"] == 2 and out["trajectory_shape"] == [2, nf, 16] and len(out["per_rank"]) == 2
    gathered = np.load(two)
    for r in range(2):
        one = tmp_path / ("one%d.npy" % r)
        o = _run([sys.executable] + common + ["--rank-offset", str(r), "--dump-trajectories", str(one)])
        assert o["per_rank"][0]["tracked_points"] == out["per_rank"][r]["tracked_points"]
        assert np.load(one)[0].tobytes() == gathered[r].tobytes(), r
    assert not np.array_equal(gathered[0], gathered[1])    # the ranks really ran different sequences


def test_round_robin_batch_equals_single_rank_run(tmp_path):
    """The fixed-batch workload (BASELINE configs[4] at a reduced frame size): six frame pairs sharded
    round-robin over two ranks give, pair by pair, the very results of one rank running all six."""
    common = ["bench.py", "--workload", "batch", "--batch-pairs", "6", "--width", "640", "--height", "360", "--features",
              "400", "--steps", "2", "--warmup", "1"]
    two, one = tmp_path / "two.npy", tmp_path / "one.npy"
    o2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", "29643"] + common + ["--gpus", "2", "--dump-pairs", str(two)],
              {"KTLVO_BENCH_REHEARSAL": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    o1 = _run([sys.executable] + common + ["--dump-pairs", str(one)])
    assert o2["scaling"] == "strong" and o2["n_gpus"] == 2 and o2["config"]["pairs_per_rank"] == [3, 3]
    assert o1["config"]["pairs_per_rank"] == [6]
    a, b = np.load(two), np.load(one)
    assert a.shape == (6, 5) and a.tobytes() == b.tobytes()
    assert o1["pair_table_crc32"] == o2["pair_table_crc32"]
    assert (a[:, 2] > 200).all() and len(set(a[:, 4].tolist())) == 6     # every pair tracked, every pair different


def test_batch_smaller_than_the_rank_count(tmp_path):
    """Two frame pairs dealt to three ranks: the third owns nothing, builds nothing, launches nothing, and still takes
    part in the barrier, the all_gather and the all_reduce; the table is the single-rank table."""
    common = ["bench.py", "--workload", "batch", "--batch-pairs", "2", "--width", "640", "--height", "360", "--features",
              "400", "--steps", "2", "--warmup", "1"]
    three, one = tmp_path / "three.npy", tmp_path / "one.npy"
    o3 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr",
               "127.0.0.1", "--master-port", "29645"] + common + ["--gpus", "3", "--dump-pairs", str(three)],
              {"KTLVO_BENCH_REHEARSAL": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    o1 = _run([sys.executable] + common + ["--dump-pairs", str(one)])
    assert o3["n_gpus"] == 3 and o3["config"]["pairs_per_rank"] == [1, 1, 0]
    assert o3["per_rank"][2]["features_in"] == 0 and o3["per_rank"][2]["tracked"] == 0
    assert np.load(three).tobytes() == np.load(one).tobytes() and o3["pair_table_crc32"] == o1["pair_table_crc32"]
    assert o3["value"] > 0


def test_sequence_workload_at_the_benchmark_frame_size(tmp_path):
    """BASELINE configs[0]/[3] name 1241x376 frames: the sequence workload takes the size from the command line."""
    o = _run([sys.executable, "bench.py", "--workload", "sequence", "--frames", "6", "--width", "1241", "--height", "376"])
    assert "1241x376" in o["metric"] and o["per_rank"][0]["tracked_points"] > 500
    assert o["per_rank"][0]["ate_rmse_m"] < 0.1
