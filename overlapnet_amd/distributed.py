"""One-vs-N candidate sweep sharded over the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Every candidate is an independent unit against the one query (no cross-candidate term anywhere in the
reference's heads, generateNet.py:64-116,327-354), so the shards never exchange data on the compute
path.  The only collective is the final gather of 8 bytes per candidate -- (overlap f32, yaw i32) -- to
the root, issued once per query.  The reference has no distributed code; this is the multi-GPU form of
`Infer.infer_multiple` (infer.py:162-203).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


SLOT_ALIGN = 32   # the Delta kernels' summation order is a function of (candidate slot mod 32): csrc/delta_head_f16x3.hip


def shard_bounds(n: int, world: int, rank: int, align: int = 1) -> Tuple[int, int]:
    """Contiguous block partition: the first (n mod world) ranks get one extra candidate.  align > 1: blocks of `align` candidates
    are dealt out instead (every shard starts at a multiple of `align`; the last block may be short).  With align = SLOT_ALIGN a
    candidate keeps its pool slot modulo 32 inside its shard, which makes the sharded sweep reproduce the unsharded one bit for
    bit (the kernels' summation order follows that slot)."""
    if world <= 0 or not (0 <= rank < world) or align < 1:
        raise ValueError("bad rank/world")
    n = int(n)
    if align > 1:
        lo, hi = shard_bounds((n + align - 1) // align, world, rank)
        return min(lo * align, n), min(hi * align, n)
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard_sizes(n: int, world: int, align: int = 1) -> List[int]:
    return [shard_bounds(n, world, r, align)[1] - shard_bounds(n, world, r, align)[0] for r in range(world)]


# ---- ownership of a GROWING cache (Infer.infer_multiple caches one frame per call, infer.py:184-185): skewed block-cyclic ------
def frame_owner(frame_id, world: int, block: int = SLOT_ALIGN):
    """Rank that keeps frame `frame_id`'s feature volume / spectrum / Delta row: `(f // block + f % block) % world`.
    Contiguous shards cannot be used for a cache that grows by one frame per query (their bounds would move), and the Delta kernels'
    summation order needs every frame at a local slot congruent to its id modulo 32 (`frame_slot`).  Dealing whole blocks of 32
    consecutive frames to one rank (rounds 1-4) satisfies that but puts a GATED reference list -- demo3_lcd.py:92-115 keeps a window
    of consecutive frame ids -- on one or two ranks; the skew spreads consecutive frames over consecutive ranks and keeps the slot
    residue: within a round of `world` blocks a rank sees every residue f % block exactly once.  Works on ints and integer arrays."""
    if np.isscalar(frame_id):
        f = int(frame_id)
        return (f // block + f % block) % world
    f = np.asarray(frame_id)
    return (f // block + f % block) % world


def frame_slot(frame_id, world: int, block: int = SLOT_ALIGN):
    """Index of frame `frame_id` in its owner's local cache: round (of `world` blocks) * block + f % block -- congruent to the frame
    id modulo `block`, so the pair (frame, query) gets the unsharded sweep's bits.  Slots of a rank are NOT filled in increasing
    order (frame 8 lands on rank 0's slot 8 long before frame 39 fills its slot 7 at world = 8): the cache is slot-addressed."""
    f = np.asarray(frame_id) if not np.isscalar(frame_id) else int(frame_id)
    return (f // (block * world)) * block + f % block


def local_capacity(n_frames: int, world: int, block: int = SLOT_ALIGN) -> int:
    """Slots a rank's cache needs once frames 0 .. n_frames - 1 have been fed (upper bound: whole rounds)."""
    return ((int(n_frames) + block * world - 1) // (block * world)) * block


def share_imbalance(frame_ids, world: int, block: int = SLOT_ALIGN) -> float:
    """max / mean of the per-rank share of a reference list (1.0 = perfectly balanced; `world` = everything on one rank)."""
    f = np.asarray(frame_ids).reshape(-1)
    if f.size == 0:
        return 1.0
    counts = np.bincount(frame_owner(f, world, block), minlength=world)
    return float(counts.max() * world / f.size)


def pack_scores(overlap: torch.Tensor, yaw: torch.Tensor) -> torch.Tensor:
    """(n) f32 + (n) i32 -> (n, 2) i32 payload (bit-cast, exact) so ONE collective moves both."""
    return torch.stack([overlap.contiguous().view(torch.int32), yaw.to(torch.int32)], dim=1).contiguous()


def unpack_scores(buf: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    return buf[:, 0].contiguous().view(torch.float32), buf[:, 1].contiguous()


def gather_scores(overlap: torch.Tensor, yaw: torch.Tensor, n_total: int, group=None, dst: int = 0, align: int = 1
                  ) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
    """Gather the per-rank (overlap, yaw) shards, in candidate order, on rank `dst` (None elsewhere).
    A single fixed-size collective: shards are padded to the largest shard size."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_total, world, align)
    if overlap.numel() != sizes[rank]:
        raise ValueError("rank %d holds %d scores, its shard has %d" % (rank, overlap.numel(), sizes[rank]))
    m = max(sizes) if sizes else 0
    payload = torch.zeros((m, 2), dtype=torch.int32, device=overlap.device)
    if overlap.numel():
        payload[:overlap.numel()] = pack_scores(overlap, yaw)
    payload = _comm_device(payload, group)      # RCCL gathers device tensors, gloo host tensors
    if rank == dst:
        bufs = [torch.empty_like(payload) for _ in range(world)]
        dist.gather(payload, bufs, dst=dst, group=group)
        cat = torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)
        return unpack_scores(cat)
    dist.gather(payload, None, dst=dst, group=group)
    return None


def sweep_one_vs_n(score_fn: Callable[[int, int], Tuple[torch.Tensor, torch.Tensor]], n_total: int, group=None,
                   dst: int = 0, align: int = 1):
    """Run `score_fn(lo, hi)` -> (overlap, yaw) on this rank's block of the candidate pool and gather."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n_total, world, rank, align)
    ov, yw = score_fn(lo, hi)
    return gather_scores(ov, yw, n_total, group, dst, align)


def _comm_device(t: torch.Tensor, group=None) -> torch.Tensor:
    """gloo moves host tensors, nccl (= RCCL) device tensors."""
    return t.cpu() if dist.get_backend(group) == "gloo" else t


def allgather_by_owner(overlap: torch.Tensor, yaw: torch.Tensor, owner: np.ndarray, group=None, status: int = 0
                       ) -> Tuple[torch.Tensor, torch.Tensor, np.ndarray]:
    """Every rank holds the (overlap, yaw) of the list entries it owns (`owner[i]` = rank of entry i, known to all ranks), in list
    order; ONE all-gather of 8 B per entry (padded to the largest share, + one status row) gives every rank the whole list in list
    order and every rank's `status` word (0 = fine): a rank whose local work failed still takes part in the collective (with
    whatever it holds, or nothing) and ALL ranks learn about it from the same payload, instead of the others blocking in a
    collective the failed rank never enters.  Returns (overlap, yaw, statuses (world,))."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    owner = np.asarray(owner)
    counts = np.bincount(owner, minlength=world).astype(np.int64) if owner.size else np.zeros(world, np.int64)
    if status == 0 and overlap.numel() != counts[rank]:
        raise ValueError("rank %d holds %d scores, it owns %d list entries" % (rank, overlap.numel(), counts[rank]))
    m = int(counts.max()) if counts.size else 0
    payload = torch.zeros((m + 1, 2), dtype=torch.int32, device=overlap.device)
    if status == 0 and overlap.numel():
        payload[:overlap.numel()] = pack_scores(overlap, yaw)
    payload[m, 0] = int(status)
    payload = _comm_device(payload, group)
    bufs = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(bufs, payload, group=group)
    statuses = torch.stack([b[m, 0] for b in bufs]).cpu().numpy()
    # list position of the k-th entry of rank r's share = the k-th position whose owner is r: one stable sort, one indexed store
    order = torch.from_numpy(np.argsort(owner, kind="stable")).to(payload.device)
    out = torch.zeros((len(owner), 2), dtype=torch.int32, device=payload.device)
    if len(owner):
        out[order] = torch.cat([bufs[r][:int(counts[r])] for r in range(world)], dim=0)
    ov, yw = unpack_scores(out)
    return ov, yw, statuses


def best_match(overlap: torch.Tensor, yaw: torch.Tensor, threshold: float = 0.3):
    """Loop-closure decision of demo3 (demo3_lcd.py:118-120): argmax overlap if it exceeds the threshold."""
    if overlap.numel() == 0:
        return None
    i = int(torch.argmax(overlap))
    if float(overlap[i]) > threshold:
        return i, float(overlap[i]), int(yaw[i])
    return None


def merge_matches(records) -> "torch.Tensor":
    """Best of per-shard best-match records ((world, 4) int32: {id, float bits of overlap, yaw, found}).
    Shards are contiguous and ordered by rank, and each record already holds its shard's first maximum, so
    'largest overlap, lowest rank on ties' reproduces np.argmax over the whole pool (demo3_lcd.py:119-120).
    Empty shards carry id -1."""
    rec = records.reshape(-1, 4).to(torch.int32).cpu()
    best = None
    for r in rec:
        if int(r[0]) < 0:
            continue
        v = float(r[1:2].view(torch.float32)[0])
        if best is None or v > best[0]:
            best = (v, r)
    if best is None:
        return torch.tensor([-1, 0, 0, 0], dtype=torch.int32)
    return best[1].clone()


def merge_matches_by_position(records) -> "torch.Tensor":
    """Best of per-rank records whose id field is the POSITION of the candidate in the caller's reference list (shares of a list
    that are not contiguous, `frame_owner`): largest overlap, lowest position on ties == np.argmax over the whole list
    (demo3_lcd.py:119-120)."""
    rec = records.reshape(-1, 4).to(torch.int32).cpu()
    best = None
    for r in rec:
        if int(r[0]) < 0 or int(r[3]) == 0:
            continue
        v = float(r[1:2].view(torch.float32)[0])
        if best is None or v > best[0] or (v == best[0] and int(r[0]) < int(best[1][0])):
            best = (v, r)
    if best is None:
        return torch.tensor([-1, 0, 0, 0], dtype=torch.int32)
    return best[1].clone()


def allgather_records(record: torch.Tensor, group=None) -> torch.Tensor:
    """(world, 4) int32: every rank's 16-byte best-match record.  The `found` field (word 3) of a rank whose local work failed
    carries a negative status instead (see `Infer._infer_best_match_sharded`): every rank sees it in the same payload."""
    world = dist.get_world_size(group)
    rec = _comm_device(record.contiguous(), group)
    bufs = [torch.empty_like(rec) for _ in range(world)]
    dist.all_gather(bufs, rec, group=group)
    return torch.stack([b.cpu() for b in bufs])


def best_match_sharded(record: torch.Tensor, group=None) -> "torch.Tensor":
    """All-gather the 16-byte per-rank records of `OvnEngine.best_match` (RCCL on GPU tensors, gloo on CPU tensors)
    and merge: every rank gets the global decision; only world x 16 bytes cross xGMI instead of N scores."""
    world = dist.get_world_size(group)
    bufs = [torch.empty_like(record) for _ in range(world)]
    dist.all_gather(bufs, record.contiguous(), group=group)
    return merge_matches(torch.stack([b.cpu() for b in bufs]))
