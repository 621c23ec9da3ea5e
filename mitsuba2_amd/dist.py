"""Multi-GPU plumbing: one process per GPU, pixel-tile (or sample-pass) sharding, one film reduce.

The path shards by pixels only (a pixel's spp samples share one PCG32 stream,
src/librender/integrator.cpp:196-209): rank r renders the spiral blocks with
id % world == r into a private full-size film (so the 2-pixel filter border of a
block lands in neighbouring blocks' texels without communication, like the
reference's bordered ImageBlock + Film::put, imageblock.cpp:49-77) and ONE
reduce(sum) of the film closes the render. backend "nccl" is RCCL on ROCm (over
xGMI); "gloo" is used by the CPU tests.
"""
import os


def init(backend=None):
    """-> (rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if not dist.is_initialized():
            dist.init_process_group(backend or "nccl", rank=rank, world_size=world)
    return rank, world, local_rank


def shard_blocks(n_blocks, rank, world):
    """Spiral block ids rendered by `rank` (must match PathIntegrator::set_shard in the host layer)."""
    return list(range(rank, n_blocks, world))


def choose_shard(mode, parts, width, height, spp):
    """How `parts` ranks split a width x height x spp frame -> "tiles" | "passes".
    tiles (the north star's partition, and what "auto" always picks): spiral blocks dealt round-robin (shard_blocks),
    every rank renders all spp of its pixels; the N-GPU film equals the 1-GPU film (to the float32 association of the
    <= 4 block partials under a block border). passes (only on explicit request): the reference's samples_per_pass =
    spp / parts run (integrator.cpp:75-86, spiral.cpp:41), pass r on rank r — every rank keeps all pixels, but the
    film is the one scalar_rgb produces for that samples_per_pass, i.e. other random numbers than the 1-GPU job's:
    a different workload, reported as such (config.parallelism), never chosen silently. Passes need spp divisible
    by parts; otherwise (and for a single rank) the answer is tiles."""
    if mode not in ("auto", "tiles", "passes"):
        raise ValueError("shard mode must be auto, tiles or passes")
    if parts <= 1 or spp % parts or mode != "passes":
        return "tiles"
    return "passes"


def pass_job(make_integrator, sensor, rank, parts, spp):
    """The render job of rank `rank` under pass sharding: pass `rank` of the samples_per_pass = spp / parts run, rendered
    into the rank's own film (accumulate = 0: the film reduce adds the passes). -> (integrator, job)"""
    integ = make_integrator(samples_per_pass=spp // parts)
    job = integ.render_job(sensor, pass_index=rank)
    job.cfg.accumulate = 0
    return integ, job


def reduce_film(film, dst=0):
    """Sum the per-rank partial films onto rank `dst` (in place). film: torch tensor, float32/float64."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value


def finalize():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
