"""Multi-sequence / multi-GPU driver logic.

The path shards across SEQUENCES only: frame t of a sequence needs the pose,
points and filter states of frame t-1 (src/lib/stereo_slam.cpp:125,183-196),
so a sequence never leaves its GPU and there is NO data-path collective.
One process per GPU owns `seqs_per_rank` sequences (one svo_ctx, sequence =
grid dimension); ranks only meet at the barriers around the timed region and
in one small all_gather of per-sequence summaries at the end (RCCL over xGMI
on the GPU box — backend "nccl" — or gloo in the CPU tests).
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend=None):
    """Process group from the torchrun environment; single process when WORLD_SIZE <= 1."""
    rank, local_rank, world = rank_info()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def sequence_ids(rank, world, seqs_per_rank):
    """Global ids (= scene seeds) of the sequences owned by `rank`: weak scaling,
    every rank owns the same number of independent sequences."""
    return list(range(rank * seqs_per_rank, (rank + 1) * seqs_per_rank))


def assign_longest_first(lengths, n_ranks):
    """Static longest-first assignment of whole sequences to ranks (a sequence cannot be
    split, frame t needs frame t-1): sequences by decreasing length, each to the rank with the least
    work so far. Returns one list of sequence indices per rank (ranks may stay empty: six EuRoC
    sequences on eight GPUs keep six busy)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    load = [0] * n_ranks
    out = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += lengths[i]
    return out


def play_unequal(slam, frames_of, lengths, time_of=lambda k: k / 20.0):
    """Drive the sequences of one ctx to their own ends: at step k every sequence that still has a
    frame gets it, the others sit the step out (svo_new_images with NULL pointers). frames_of(s, k)
    -> (left, right). Returns the number of sequence-frames processed."""
    done = 0
    for k in range(max(lengths) if lengths else 0):
        L, R = [], []
        for s, n in enumerate(lengths):
            if k < n:
                l, r = frames_of(s, k)
                L.append(l); R.append(r)
                done += 1
            else:
                L.append(None); R.append(None)
        slam.new_images(L, R, [time_of(k)] * len(lengths))
    return done


def _sync(device):
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def _barrier(world, device):
    if world > 1:
        if device is not None and device.type == "cuda":
            dist.barrier(device_ids=[device.index])
        else:
            dist.barrier()


def timed_steps(step_fn, steps, warmup, world, device=None, coll_device="same", finish_fn=None):
    """Driver contract: `warmup` untimed steps, then exactly `steps` steps bracketed
    by barrier + device synchronize on both sides; returns MAX-over-ranks seconds.
    `device` is synchronized; collectives run on `coll_device` (default: the same;
    None = host tensors, e.g. gloo). `finish_fn` drains work that step_fn only queued
    (svo_submit_images): it runs inside the timed region, before the closing synchronize."""
    if coll_device == "same":
        coll_device = device
    for k in range(warmup):
        step_fn(k)
    if finish_fn:
        finish_fn()
    _sync(device)
    _barrier(world, coll_device)
    _sync(device)
    t0 = time.perf_counter()
    for k in range(warmup, warmup + steps):
        step_fn(k)
    if finish_fn:
        finish_fn()
    _sync(device)
    t1 = time.perf_counter()
    _barrier(world, coll_device)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64,
                           device=coll_device if (coll_device is not None and coll_device.type == "cuda") else "cpu")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    return float(elapsed.item())


def gather_summaries(local, world, device=None):
    """all_gather of the fixed-size per-sequence summaries ([n_local, k] float64):
    the one exchange of the multi-GPU path. Returns [world * n_local, k]."""
    t = torch.as_tensor(np.asarray(local, np.float64))
    if world <= 1:
        return t.numpy()
    if device is not None and device.type == "cuda":
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return torch.cat(out, 0).cpu().numpy()


def throughput(total_units, seconds):
    return total_units / seconds if seconds > 0 else float("nan")
