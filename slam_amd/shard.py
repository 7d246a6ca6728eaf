"""Multi-GPU partitioning of the feature-tracking path.

Features, frame pairs and sequences are independent, so work is sharded across ranks with no
data-path collective; the only exchange is one all_gather of a fixed-size per-rank record
(statistics and, for sequence runs, the 4x4 poses) at the end.  Backend "nccl" is RCCL on ROCm;
"gloo" runs the same code on CPU for tests.
"""
import numpy as np
import torch
import torch.distributed as dist

RECORD_FIELDS = ("features_in", "tracked", "pairs", "k3_ms", "k3_launches", "k1_ms", "k1_launches", "elapsed_s")


def rank_seed(base_seed, rank):
    """Each rank renders its own sequence: seed + rank (one sequence per GPU)."""
    return (base_seed + rank) & 0xFFFFFFFF


def shard_round_robin(n_items, rank, world):
    """Indices of the items (frame pairs, sequences) owned by `rank`."""
    return list(range(rank, n_items, world))


def _single(world):
    """No process group (a plain `python bench.py` run): nothing to exchange.  Under torchrun the group
    exists even for one rank, and the collectives run (so a one-GPU box exercises the RCCL path too)."""
    return world == 1 and not (dist.is_available() and dist.is_initialized())


def _collective_tensor(t):
    """RCCL ("nccl") exchanges device tensors; gloo (CPU tests, rehearsals) wants host tensors."""
    return t.cpu() if dist.get_backend() == "gloo" else t


def gather_records(rec, world):
    """all_gather of one fixed-size float64 record per rank -> array [world, len(rec)]."""
    if _single(world):
        return rec.detach().cpu().numpy()[None].copy()
    rec = _collective_tensor(rec)
    out = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    return torch.stack(out).cpu().numpy()


def gather_trajectories(poses, world):
    """poses: float32 [frames, 16] on this rank -> [world, frames, 16] (same frame count everywhere)."""
    if _single(world):
        return poses.detach().cpu().numpy()[None].copy()
    poses = _collective_tensor(poses)
    out = [torch.zeros_like(poses) for _ in range(world)]
    dist.all_gather(out, poses)
    return torch.stack(out).cpu().numpy()


PAIR_FIELDS = ("pair", "features_in", "tracked", "status_crc32", "positions_crc32")


def gather_pair_records(rows, n_items, world):
    """Per-item results of a round-robin sharded batch, gathered and put back in item order.

    rows: int64 [k, len(PAIR_FIELDS)] for the k items this rank owns (column 0 = item index).  Every rank
    contributes a block padded to ceil(n_items / world) rows (item index -1 marks padding), so the one
    all_gather moves fixed-size blocks.  Returns int64 [n_items, len(PAIR_FIELDS)] ordered by item index,
    the same array whatever the number of ranks."""
    width = len(PAIR_FIELDS)
    per_rank = -(-n_items // world)
    block = torch.full((per_rank, width), -1, dtype=torch.int64, device=rows.device)
    block[: rows.shape[0]] = rows
    if _single(world):
        blocks = [block.cpu()]
    else:
        block = _collective_tensor(block)
        out = [torch.zeros_like(block) for _ in range(world)]
        dist.all_gather(out, block)
        blocks = [b.cpu() for b in out]
    table = np.full((n_items, width), -1, np.int64)
    for b in blocks:
        for row in b.numpy():
            if row[0] >= 0:
                assert table[row[0], 0] == -1, "item %d reported twice" % row[0]
                table[row[0]] = row
    assert (table[:, 0] == np.arange(n_items)).all(), "an item of the batch was not reported"
    return table


def max_over_ranks(value, device, world):
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if not _single(world):
        t = _collective_tensor(t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate(recs, steps, t_max):
    """Whole-job numbers from the gathered records (rank 0 prints these)."""
    recs = np.asarray(recs, dtype=np.float64)
    total_tracked = float(recs[:, 1].sum())
    return {
        "features_in_per_step": float(recs[:, 0].sum()),
        "tracked_per_step": total_tracked,
        "value": total_tracked * steps / t_max,
        "pairs_per_s": float(recs[:, 2].sum()) * steps / t_max,
    }
