"""CPU tests of the host-side logic: synthetic data generator, work partitioning and the N > 1
exchange (gloo, world size 2) that bench.py runs over RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slam_amd import shard, synth


def test_texture_and_scene_are_deterministic():
    a1 = synth.texture_image(160, 120, seed=3, dx=0.25, dy=-0.5)
    a2 = synth.texture_image(160, 120, seed=3, dx=0.25, dy=-0.5)
    assert np.array_equal(a1, a2) and a1.dtype == np.uint8 and 20 < a1.std() < 70
    assert not np.array_equal(a1, synth.texture_image(160, 120, seed=4, dx=0.25, dy=-0.5))
    s = synth.CorridorScene(320, 96, seed=9)
    l1, r1 = s.stereo_pair(2)
    l2, r2 = synth.CorridorScene(320, 96, seed=9).stereo_pair(2)
    assert np.array_equal(l1, l2) and np.array_equal(r1, r2) and not np.array_equal(l1, r1)


def test_scene_geometry_matches_its_ground_truth():
    s = synth.CorridorScene(1241, 376, seed=0x5EED0001)
    assert abs(s.bf - 0.5461 * 707.0912) < 1e-9 and abs(s.cx - (601.8873 - 0.5)) < 1e-9
    img, depth = s.render(0, with_depth=True, noise=False)
    assert depth.min() > 0 and depth[300, 620] < depth[200, 620]          # ground plane recedes upwards
    # a ground pixel seen by the left camera reappears in the right image shifted by bf / depth
    v, u = 330, 500
    d = s.bf / depth[v, u]
    right = s.render(0, right=True, noise=False)
    patch = img[v - 4:v + 5, u - 4:u + 5].astype(np.int32)
    best = min(range(-3, 4), key=lambda k: np.abs(
        right[v - 4:v + 5, int(round(u - d)) + k - 4:int(round(u - d)) + k + 5].astype(np.int32) - patch).sum())
    assert abs(best) <= 1
    T = s.relative_pose(0, 1)
    assert abs(T[2, 3] + 0.8) < 0.05 and np.abs(T[:3, :3] @ T[:3, :3].T - np.eye(3)).max() < 1e-12


def test_partitioning_helpers():
    assert shard.shard_round_robin(10, 1, 4) == [1, 5, 9]
    got = sorted(i for r in range(8) for i in shard.shard_round_robin(64, r, 8))
    assert got == list(range(64))
    assert shard.rank_seed(0x5EED0010, 3) == 0x5EED0013
    recs = np.array([[100, 90, 2, 1, 1, 1, 1, 0.5], [100, 80, 2, 1, 1, 1, 1, 0.6]])
    agg = shard.aggregate(recs, steps=3, t_max=0.6)
    assert agg["tracked_per_step"] == 170 and abs(agg["value"] - 170 * 3 / 0.6) < 1e-9


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    rec = torch.tensor([2000.0 * (rank + 1), 1900.0 + rank, 32.0, 0.5, 20, 0.04, 20, 0.1 + 0.05 * rank],
                       dtype=torch.float64)
    recs = shard.gather_records(rec, world)
    t_max = shard.max_over_ranks(0.1 + 0.05 * rank, dev, world)
    poses = torch.full((5, 16), float(rank), dtype=torch.float32)
    traj = shard.gather_trajectories(poses, world)
    dist.barrier()
    if rank == 0:
        q.put((recs, t_max, traj))
    dist.destroy_process_group()


def _pair_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_items = 7                                            # not a multiple of the rank count: one block is padded
    mine = shard.shard_round_robin(n_items, rank, world)
    rows = torch.tensor([[p, 100, 90 + p, 1000 + p, 2000 + p] for p in mine], dtype=torch.int64).reshape(-1, 5)
    table = shard.gather_pair_records(rows, n_items, world)
    dist.barrier()
    if rank == 0:
        q.put(table)
    dist.destroy_process_group()


def test_round_robin_batch_gather_world_size_2_gloo():
    """The fixed-batch workload's exchange: per-pair rows from two ranks come back in pair order and equal
    the table a single rank produces for the same batch."""
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_pair_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    table = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rows = torch.tensor([[p, 100, 90 + p, 1000 + p, 2000 + p] for p in range(7)], dtype=torch.int64)
    single = shard.gather_pair_records(rows, 7, 1)
    assert table.shape == (7, 5) and np.array_equal(table, single)
    assert list(table[:, 0]) == list(range(7))


def test_gather_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    recs, t_max, traj = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert recs.shape == (2, 8)
    assert list(recs[:, 0]) == [2000.0, 4000.0] and list(recs[:, 1]) == [1900.0, 1901.0]
    assert abs(t_max - 0.15) < 1e-12                                   # max over ranks, as the bench reports
    assert traj.shape == (2, 5, 16) and (traj[0] == 0).all() and (traj[1] == 1).all()
    # gathered results equal what two single-process runs produce
    single = [shard.gather_records(torch.tensor(recs[r]), 1)[0] for r in range(2)]
    assert np.array_equal(np.stack(single), recs)


def _pair_worker_n(rank, world, port, q, n_items):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_round_robin(n_items, rank, world)
    rows = torch.tensor([[p, 100, 90 + p, 1000 + p, 2000 + p] for p in mine], dtype=torch.int64).reshape(-1, 5)
    table = shard.gather_pair_records(rows, n_items, world)
    # a rank without pairs still reports: zero features, zero tracked, its own elapsed time
    rec = torch.tensor([100.0 * len(mine), 90.0 * len(mine), float(len(mine)), 0.5 * len(mine), float(len(mine) > 0), 0.0, 0.0,
                        0.01 * (rank + 1)], dtype=torch.float64)
    recs = shard.gather_records(rec, world)
    t_max = shard.max_over_ranks(0.01 * (rank + 1), torch.device("cpu"), world)
    traj = shard.gather_trajectories(torch.full((3, 16), float(rank), dtype=torch.float32), world)
    dist.barrier()
    if rank == 0:
        q.put((table, recs, t_max, traj))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [6, 8, 19])
def test_gathers_world_size_8_gloo(n_items):
    """Eight ranks, as the 8-GPU node runs them: a batch smaller than the rank count (6 pairs: two ranks own nothing
    and contribute an all-padding block and a zero record), one pair each, and an uneven deal.  The pair table equals
    the single-rank table, the per-rank records and trajectories arrive in rank order, the time is the slowest rank's."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_pair_worker_n, args=(r, world, port, q, n_items)) for r in range(world)]
    for p in procs:
        p.start()
    table, recs, t_max, traj = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    rows = torch.tensor([[p, 100, 90 + p, 1000 + p, 2000 + p] for p in range(n_items)], dtype=torch.int64)
    assert np.array_equal(table, shard.gather_pair_records(rows, n_items, 1))
    owned = [len(shard.shard_round_robin(n_items, r, world)) for r in range(world)]
    assert sum(owned) == n_items and (min(owned) == 0) == (n_items < world)
    assert recs.shape == (world, 8) and list(recs[:, 2]) == [float(k) for k in owned]
    assert abs(t_max - 0.08) < 1e-12
    agg = shard.aggregate(recs, 10, t_max)
    assert agg["tracked_per_step"] == 90.0 * n_items and abs(agg["value"] - 90.0 * n_items * 10 / 0.08) < 1e-6
    assert traj.shape == (world, 3, 16) and all((traj[r] == r).all() for r in range(world))


def test_avx2_build_of_the_oracle_equals_the_checker_build():
    """bench.py times a build of the oracle made for AVX2-class CPUs; with -ffp-contract=off in both builds its
    results must be the checker's bit for bit, in every reading of the window sums, on any number of threads."""
    import oracle as O
    from slam_amd import synth
    a, b = synth.texture_pair(320, 240, 5, 1.5, -0.5, noise=True)
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(-10, 330, 300), rng.uniform(-10, 250, 300)], 1).astype(np.float32)
    base = O.Baseline()
    for sums in (O.SUMS_SERIAL, O.SUMS_FOUR_LANES, O.SUMS_EXACT):
        ref = O.pyrlk(a, b, pts, sums=sums)
        for threads in (1, 3):
            got = base.track_pair(a, b, pts, threads=threads, sums=sums)
            assert all(np.array_equal(x, y) for x, y in zip(got, ref)), (sums, threads)


def test_parallel_frame_renderer_matches_serial():
    """bench.py renders its input frames in helper processes; they must be the serial renderer's."""
    from slam_amd import synth
    frames = synth.render_corridor_frames(160, 96, 0x5EED0002, 5, workers=3)
    scene = synth.CorridorScene(160, 96, seed=0x5EED0002)
    assert len(frames) == 5
    for f, img in enumerate(frames):
        assert img.dtype == np.uint8 and img.shape == (96, 160)
        assert np.array_equal(img, scene.render(f))


def test_bench_cpu_baseline_leg_runs_on_the_host():
    """bench.py's CPU leg (the oracle timed on a bounded sample) needs no GPU; run it on two tiny frames."""
    import importlib.util
    import os
    import numpy as np
    from slam_amd import synth
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for name in ("main", "parse", "run_sequence", "run_ransac", "run_batch", "cpu_baseline", "measured_traffic"):
        assert callable(getattr(bench, name)), name
    a, b = synth.texture_pair(320, 240, 5, 1.5, -0.5, noise=True)
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(30, 290, 64), rng.uniform(30, 210, 64)], 1).astype(np.float32)[None]
    out = bench.cpu_baseline([a, b], pts, 4, 0.2)
    assert out["kind"] == "port" and out["value"] > 0 and out["cores"] >= 1
    assert len(out["lk_iterations_per_level_mean"]) >= 3 and out["single_thread"]["value"] > 0
    assert out["four_accumulators"]["value"] > 0 and out["all_cores_over_one_thread"] > 0 and "-O3" in out["build"]
    assert out["pairs_in_parallel"]["value"] > 0 and out["pairs_in_parallel_over_one_thread"] > 0
    assert out["value"] == max(out["pairs_in_parallel"]["value"], out["stripes_within_a_pair"]["value"])
    assert out["arrangement"] in ("pairs_in_parallel", "stripes_within_a_pair")
