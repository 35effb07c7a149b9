"""Batch sharding across GPUs: independent proofs of a batch go to rank i mod world; the only
data that crosses xGMI is the finished proof bytes, moved with one all_gather per step (RCCL on GPUs: backend "nccl";
the CPU tests run the same code over gloo)."""
import torch


def shard_indices(num_proofs: int, rank: int, world: int):
    """Proof indices owned by `rank`: round-robin, so any batch size spreads evenly."""
    return list(range(rank, num_proofs, world))


def gather_proofs(local: dict, num_proofs: int, proof_len: int, dist=None, device="cpu", force_collective=False):
    """local: {proof index: proof bytes} produced by this rank. Returns the full list (index order) on every rank.
    force_collective: run the all_gather even in a world of one (the RCCL path on a one-GPU box, tests/test_batch_gpu.py)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return [local[i] for i in range(num_proofs)]
    world, rank = dist.get_world_size(), dist.get_rank()
    per_rank = (num_proofs + world - 1) // world
    mine = torch.zeros(per_rank * proof_len, dtype=torch.uint8)
    for slot, idx in enumerate(shard_indices(num_proofs, rank, world)):
        if len(local[idx]) != proof_len:
            raise ValueError("proof %d has %d bytes, expected %d" % (idx, len(local[idx]), proof_len))
        mine[slot * proof_len:(slot + 1) * proof_len] = torch.frombuffer(bytearray(local[idx]), dtype=torch.uint8)
    mine = mine.to(device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    proofs = [None] * num_proofs
    for r in range(world):
        buf = out[r].cpu().numpy().tobytes()
        for slot, idx in enumerate(shard_indices(num_proofs, r, world)):
            proofs[idx] = buf[slot * proof_len:(slot + 1) * proof_len]
    return proofs
