"""Multi-GPU: samples-per-pixel are partitioned over the ranks of one node with the
scene replicated; the only exchange is one integer sum-reduction of the
accumulators at the end of a render (RCCL over xGMI when the process group's
backend is "nccl").  Integer adds commute (saturating ones of non-negative numbers
too), and every sample's random stream is keyed by (seed, source, sample number),
so the reduced image is bit-identical for any number of ranks.

torch is plumbing here: it owns the reduction buffers and the process group.
"""
import numpy as np


U32_MAX = 0xFFFFFFFF


def allreduce_accumulators(sum4, count):
    """In-place all-reduce of the two accumulator tensors (int32 views of the u32 fixed-point sums and
    counts).  The sums saturate at 2^32 - 1, as they do while one GPU accumulates and in mirt_group_reduce:
    they travel as 64-bit integers (a sum of at most 2^31 ranks' u32 values cannot wrap there) and are
    clamped before they go back into 32 bits — saturating adds of non-negative numbers in any order end on
    min(total, 2^32 - 1), so every rank count gives the bits one GPU would.  Counts wrap, as everywhere."""
    import torch
    import torch.distributed as dist
    wide = sum4.to(torch.int64) & U32_MAX
    dist.all_reduce(wide, op=dist.ReduceOp.SUM)
    wide.clamp_(max=U32_MAX)
    sum4.copy_(torch.where(wide > 0x7FFFFFFF, wide - (1 << 32), wide).to(torch.int32))
    dist.all_reduce(count, op=dist.ReduceOp.SUM)


def reduce_device_accumulators(pt, device):
    """Combine the accumulators of every rank's Pathtracer `pt` on its GPU: copy
    them into torch tensors (device-to-device), all-reduce, load the result back.
    Returns the seconds spent in the collective (after a device synchronise)."""
    import time

    import torch

    n = pt.n_sources
    t4 = torch.empty(4 * n, dtype=torch.int32, device=device)
    tc = torch.empty(n, dtype=torch.int32, device=device)
    pt.copy_accum_to_device(t4.data_ptr(), tc.data_ptr())
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    allreduce_accumulators(t4, tc)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    pt.load_accum_from_device(t4.data_ptr(), tc.data_ptr())
    return dt


def payload_bytes(n_sources):
    """Bytes each rank contributes to the reduction (16 B sums + 4 B count per source)."""
    return int(n_sources) * 20
