"""Multi-GPU split of Integrate::render(): one process per GPU, `torch.distributed` (RCCL over xGMI on the GPU box,
gloo in the CPU tests).

The reference parallelises render() over image rows with rayon (src/integrators/base.rs:50-63 `par_bridge` over
the pixel rows; every pixel owns its RNG stream, src/sampling/sampler.rs:110-137).  Because a (pixel, sample)
pair fully determines its sample stream, the work shards with NO data-path communication in two ways:

  * sample split  — rank r renders sample indices [r*spp/N, (r+1)*spp/N) of every pixel; the partial films
                    (rgb_sum, weight_sum: plain sums) are then added with ONE all-reduce of the f64 film.
  * row split     — rank r renders a band of rows; the bands are disjoint, so the film is assembled with
                    ONE all-gather (bands padded to equal height).

Both end in exactly one collective over the film (32 MiB for 1024x1024), nothing inside the path loop.
"""
import numpy as np


def shard_samples(spp_total, rank, world):
    """Contiguous sample-index range [s0, s1) of `rank`; ranges differ by at most one sample."""
    base, rem = divmod(int(spp_total), int(world))
    s0 = rank * base + min(rank, rem)
    return s0, s0 + base + (1 if rank < rem else 0)


def shard_rows(bounds, rank, world, align=8):
    """Pixel window (x0, y0, x1, y1) of `rank`: bands of rows, cut at multiples of `align` rows (the renderer
    schedules 8x8 pixel tiles, so aligned bands waste no lanes).  A rank may get an empty band."""
    x0, y0, x1, y1 = [int(v) for v in bounds]
    n_units = -(-(y1 - y0) // align)
    base, rem = divmod(n_units, int(world))
    u0 = rank * base + min(rank, rem)
    u1 = u0 + base + (1 if rank < rem else 0)
    return x0, min(y0 + u0 * align, y1), x1, min(y0 + u1 * align, y1)


def allreduce_film(film, group=None, async_op=False):
    """sample split: film (torch tensor [H, W, 4] f64, device or CPU) summed over ranks in place.
    async_op: returns the collective's work handle instead (None for a single rank); the caller renders the next frame while
    the collective runs and calls .wait() before it touches `film` again."""
    import torch.distributed as dist
    work = None
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        work = dist.all_reduce(film, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else film


def allgather_rows(film, bounds, group=None, align=8, async_op=False):
    """row split: every rank holds a full-size film tensor in which only its own band is filled; returns the
    assembled film (on every rank).  async_op: returns (assemble, work) instead - the all-gather's handle and a function that
    builds the assembled film once work.wait() has returned (the caller renders the next frame in between)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return ((lambda: film), None) if async_op else film
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    x0, y0, x1, y1 = [int(v) for v in bounds]
    wins = [shard_rows(bounds, r, world, align) for r in range(world)]
    hmax = max(w[3] - w[1] for w in wins)
    mine = torch.zeros((hmax,) + tuple(film.shape[1:]), dtype=film.dtype, device=film.device)
    w = wins[rank]
    mine[: w[3] - w[1]] = film[w[1] - y0: w[3] - y0]
    parts = [torch.empty_like(mine) for _ in range(world)]
    work = dist.all_gather(parts, mine, group=group, async_op=async_op)

    def assemble():
        out = torch.empty_like(film)
        for r, w in enumerate(wins):
            out[w[1] - y0: w[3] - y0] = parts[r][: w[3] - w[1]]
        return out
    return (assemble, work) if async_op else assemble()


def render_distributed(integrator, mode="samples", group=None):
    """Render this rank's share with `integrator` (a WavefrontIntegrator built identically on every rank), merge
    the films with one collective, load the merged film back into the integrator, and return it as a torch tensor."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    integrator.clear()
    if mode == "samples":
        s0, s1 = shard_samples(integrator.params.spp, rank, world)
        if s1 > s0:
            integrator.render(samples=(s0, s1))
    else:
        win = shard_rows(integrator.bounds, rank, world)
        if win[3] > win[1]:
            integrator.render(window=win)
    film = torch.empty((integrator.height, integrator.width, 4), dtype=torch.float64, device="cuda")
    integrator.film(out=film)
    if world > 1:
        film = allreduce_film(film, group) if mode == "samples" else allgather_rows(film, integrator.bounds, group)
        # the collective and the copies that assemble the bands run on torch's streams; the library reads `film` on its own
        # (non-blocking) stream, which does not order against them: wait before handing the tensor over.  (Found in round 3 with
        # four ranks on one card: the tail of the assembled film was still being written when film_add read it.)
        torch.cuda.current_stream().synchronize() if film.is_cuda else None
        integrator.clear()
        integrator.film_add(film)
    return film
