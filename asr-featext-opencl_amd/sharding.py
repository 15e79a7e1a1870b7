"""Utterance sharding across the GPUs of one node.

The path shards trivially: utterances are independent, so rank r of n takes utterances
r, r+n, r+2n, ... (round-robin, as BASELINE.json configs[3] states) and runs the ordinary batch
entry points on its own device.  There is no data-path collective; ``torch.distributed`` (RCCL on
GPUs, gloo in the CPU tests) is used only to agree on totals and timings.

The reference has no multi-device code at all: its "threads" loop runs the worker sequentially over
one file queue (ASR_OCL.cpp:340-368).
"""
import numpy as np


def shard_indices(n_utt, rank, world):
    """Indices of the utterances rank `rank` owns (round-robin)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return np.arange(rank, n_utt, world, dtype=np.int64)


def shard_layout(lengths, rank, world):
    """Pack this rank's utterances back to back: returns (indices, offsets, lengths, total_samples).
    Offsets are kept even so that the aligned 32-bit PCM load path applies."""
    lengths = np.asarray(lengths, dtype=np.int64)
    idx = shard_indices(lengths.size, rank, world)
    ln = lengths[idx]
    padded = ln + (ln & 1)
    off = np.zeros(idx.size, dtype=np.int64)
    if idx.size > 1:
        off[1:] = np.cumsum(padded[:-1])
    total = int(padded.sum())
    return idx, off, ln, total


def frames_of(lengths, window_size, shift):
    """Frames per utterance, integer arithmetic (parambase.cpp:16-19)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    t = (lengths - (window_size - shift)) // shift
    return np.maximum(t, 0)


def gather_counts(local_frames, dist=None):
    """Sum of a per-rank frame count over all ranks (identity without torch.distributed)."""
    if dist is None or not dist.is_initialized():
        return int(local_frames)
    import torch
    t = torch.tensor([int(local_frames)], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def max_over_ranks(value, dist=None):
    """Max of a per-rank scalar (the bench's step time) over all ranks."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
