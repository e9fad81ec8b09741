"""Multi-GPU layout of the path: independent proofs, one commitment per GPU, roots assembled by ONE all-gather.

The reference has no distributed code ("Distributed communication backend: none").  BASELINE.json
configs[3] shards independent 2^20-row proofs one per GPU and assembles their Merkle roots with a single
all-gather (RCCL over xGMI; `gloo` in the CPU tests).  There is no data-path collective: a commitment never
needs another rank's rows.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def proofs_of_rank(n_proofs: int, rank: int, world: int):
    """Contiguous block partition of proof ids over ranks (first ranks take the remainder)."""
    base, rem = divmod(n_proofs, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def seed_of_proof(base_seed: int, proof_id: int) -> int:
    """Synthetic-input seed of one proof (seeds offset by proof / rank)."""
    return (base_seed + 0x9E3779B97F4A7C15 * (proof_id + 1)) & 0x7FFFFFFFFFFFFFFF


def cosets_of_rank(blowup: int, rank: int, world: int):
    """(first coset, count) of a rank when ONE packed commitment is sharded by coset (world must divide blowup)."""
    if blowup % world:
        raise ValueError(f"world size {world} must divide the blowup factor {blowup}")
    per = blowup // world
    return rank * per, per


def interleave_leaf_shards(gathered: torch.Tensor, world: int, trace_len: int, per_rank: int) -> torch.Tensor:
    """gathered: [world * trace_len * per_rank, 32] digests, rank-major, each shard ordered (k, local coset).
    Returns the leaves in natural order j = k * blowup + coset, blowup = world * per_rank."""
    g = gathered.view(world, trace_len, per_rank, 32)
    return g.permute(1, 0, 2, 3).reshape(world * trace_len * per_rank, 32).contiguous()


def all_gather_leaf_shards(local_leaves: torch.Tensor, trace_len: int, per_rank: int, group=None) -> torch.Tensor:
    """The one exchange of a coset-sharded commitment: all-gather of the leaf digests (32 B x R x cosets per rank),
    then interleave to natural order.  local_leaves: uint8 [trace_len * per_rank, 32]."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_leaves.clone()
    world = dist.get_world_size(group)
    if dist.get_backend(group) != "nccl" and local_leaves.is_cuda:  # gloo rehearsal: collectives on host copies
        return all_gather_leaf_shards(local_leaves.cpu(), trace_len, per_rank, group).to(local_leaves.device)
    out = torch.empty((world * local_leaves.shape[0], 32), dtype=torch.uint8, device=local_leaves.device)
    dist.all_gather_into_tensor(out, local_leaves.contiguous(), group=group)
    return interleave_leaf_shards(out, world, trace_len, per_rank)


def all_gather_roots(local_roots: torch.Tensor, group=None) -> torch.Tensor:
    """local_roots: uint8 [k, 32] on this rank's device (k equal on all ranks).  Returns [world*k, 32], rank-major:
    the one collective of the path."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_roots.clone()
    world = dist.get_world_size(group)
    if dist.get_backend(group) != "nccl" and local_roots.is_cuda:  # gloo rehearsal: collectives on host copies
        return all_gather_roots(local_roots.cpu(), group).to(local_roots.device)
    out = torch.empty((world * local_roots.shape[0], 32), dtype=torch.uint8, device=local_roots.device)
    dist.all_gather_into_tensor(out, local_roots.contiguous(), group=group)
    return out
