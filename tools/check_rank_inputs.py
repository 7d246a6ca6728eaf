#!/usr/bin/env python3
"""Checks, for ranks 0..N-1 of bench.py's default workload, that every frame of the rank's two synthetic
sequences yields the workload's corner count (possibly after bench.py's spacing fallback).  Run on a
GPU box: python tools/check_rank_inputs.py [n_ranks]"""
import os
import sys
from multiprocessing import Pool


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from slam_amd import shard, synth  # noqa: E402

W, H, P, NF, SEED, Q = 1241, 376, 127, 2000, 0x5EED0002, 2


def render(args):
    rank, q, f = args
    return synth.CorridorScene(W, H, seed=shard.rank_seed(SEED + 0x1000 * q, rank)).render(f)


def main():
    n_ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    jobs = [(r, q, f) for r in range(n_ranks) for q in range(Q) for f in range(P)]
    with Pool(min(16, os.cpu_count() or 1)) as pool:          # before anything touches the GPU
        frames = pool.map(render, jobs, chunksize=8)
    from slam_amd import lib as K
    cfg = K.default_config(width=W, height=H, num_slots=2, max_level=4, max_points=NF)
    ctx = K.Context(cfg, device=0)
    worst = {}
    for (r, q, f), img in zip(jobs, frames):
        ctx.upload_image(0, img)
        md = 10.0
        (xy, _), = ctx.detect_gftt(0, [(0, 0, W, H)], K.gftt_params(NF, 0.01, md, 3, 0, 0.0))
        while len(xy) < NF and md >= 1.0:
            md *= 0.9
            (xy, _), = ctx.detect_gftt(0, [(0, 0, W, H)], K.gftt_params(NF, 0.01, md, 3, 0, 0.0))
        assert len(xy) == NF, (r, q, f, len(xy))
        worst[r] = min(worst.get(r, 10.0), md)
    print("all %d ranks x 2 sequences x %d frames yield %d corners; smallest minDistance per rank: %s" % (n_ranks, P, NF, worst))


if __name__ == "__main__":
    main()
