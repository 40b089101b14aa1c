"""Row-band partition of one frame over the ranks of a torch.distributed job, and the gather that
rebuilds the frame on rank 0.

The reference has a single device; pixels are independent and the scene is 2 KB, so the
frame shards with no exchange during rendering.  Rank r owns the bands r, r+N, r+2N, ... of `band_rows`
rows each (rt_config.band_* in include/uob_rt.h) — interleaved rather than contiguous stripes because
the top/bottom rows of the Cornell-Box view are empty and the sphere rows are costlier.
The only collective is one gather of the finished ARGB bands to rank 0 (RCCL on GPUs, gloo in the CPU
tests).  torch is used for device memory and torch.distributed only.
"""


def band_rows_of(rank, world, height, band_rows):
    """Global row indices owned by `rank`, in the packed order of its output buffer."""
    return [y for y in range(height) if (y // band_rows) % world == rank]


def padded_rows(height, world, band_rows):
    """Rows of the equal-size stripe every rank sends: whole bands, as many as the rank with the most has (rank 0).
    Heights that are not a multiple of band_rows*world are fine: a rank renders only the rows it owns
    (rt_config_owned_rows) into the top of its padded stripe, and the padding falls beyond the frame's last row."""
    nbands = -(-height // band_rows)
    return -(-nbands // world) * band_rows


def gather_frame(stripe, world, rank, band_rows, recv=None, frame=None, dst=0, force=False, height=None, emulate=False):
    """Gather every rank's packed bands to `dst` and de-interleave them into image order.

    stripe: [padded_rows, W] integer tensor (ARGB words) on this rank, the rows it owns packed at the top.  `recv`:
    optional preallocated [world, padded_rows, W] receive buffer on `dst` (its slices are the gather list, so the bands
    land in one allocation and the de-interleave is a single strided copy).  `frame`: optional preallocated
    [padded_rows * world, W] buffer.  Returns the [height, W] frame on `dst` (a view of `frame`) and None elsewhere.
    With world == 1 the stripe is the frame and no collective runs, unless `force` (used to exercise the collective on
    one rank).  `emulate`: this ONE process plays rank `rank` of a `world`-rank job and the root at once (bench.py
    --emulate-rank): its stripe is gathered into slot `rank` of the root's receive buffer and the root's de-interleave runs
    over all `world` slots — the per-step work of the root of a real job, minus the other ranks' transfers.
    """
    import torch
    import torch.distributed as dist
    if world == 1 and not force:
        return stripe if height is None else stripe[:height]
    rows, width = stripe.shape
    if rows % band_rows:
        raise ValueError("the stripe must hold whole bands: %d rows, bands of %d (see padded_rows)" % (rows, band_rows))
    if (rank == dst or emulate) and recv is None:
        recv = torch.empty((world, rows, width), dtype=stripe.dtype, device=stripe.device)
    if emulate:
        dist.gather(stripe, [recv[rank]], dst=0)
    else:
        dist.gather(stripe, [recv[r] for r in range(world)] if rank == dst else None, dst=dst)
        if rank != dst:
            return None
    if frame is None:
        frame = torch.empty((rows * world, width), dtype=stripe.dtype, device=stripe.device)
    # [rank, band, row, x] -> [band, rank, row, x] == image order
    frame.view(-1, world, band_rows, width).copy_(recv.view(world, -1, band_rows, width).permute(1, 0, 2, 3))
    return frame if height is None else frame[:height]
