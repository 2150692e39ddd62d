"""Multi-GPU plumbing for the view-partitioned export: one process per GPU, launched by
torch.distributed.run; torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU) is used
for exactly three things: one broadcast of the volume, barriers, and a max-reduce of the timing.

Views are independent (sphere_screenshot_manager.rs:33-54 carries no state between them), so the data
path itself has no collective: rank r renders views {i : (i + i // x_div) % world == r} of the reference's view list --
interleaved, the interleave rotated by one from one row of the sphere to the next -- and reads back its own images.
"""
import os


def env_rank():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def partition_views(n_views, rank, world, x_div=None):
    """Interleaved view partition: balances the view-dependent cost (rays through the cube diagonal are 1.7x longer than
    axis-aligned ones) across ranks.  The view list is the sphere's rows of `x_div` azimuths each (y outer, x inner,
    sphere_screenshot_manager.rs:23-54); a plain i % world gives a rank the SAME azimuths in every row whenever world divides
    x_div -- 16 and 8: rank 0 renders only the axis-aligned columns; the slowest rank's share of the C3 export measured 5.8 %
    above the mean at 8 and at 4 ranks (tools/partition_balance.py, profiles/r3/partition_balance.txt) -- so the interleave is
    rotated by one per row: view i goes to rank (i + i // x_div) % world (0.3 % and 0.1 %).  Without x_div: plain interleave."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if not x_div:
        return list(range(rank, n_views, world))
    return [i for i in range(n_views) if (i + i // int(x_div)) % world == rank]


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world == 1 and "MASTER_ADDR" not in os.environ:
        return None
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist


def broadcast_volume(tensor, src=0):
    """the one collective of the path: the RGBA8 volume from rank `src` to every rank, in place."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(tensor, src=src)
    return tensor


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value, device="cpu"):
    """every rank's value, in rank order, on every rank (the per-rank spread of the bench line)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    world, rank = dist.get_world_size(), dist.get_rank()
    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[rank] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]
