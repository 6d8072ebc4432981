"""Track-sharded batches across the GPUs of one node.

Tracks are independent units (the reference processes one file per call, `src/audio_cut/api.py:102-109`,
and its only multi-device tool loops devices sequentially, `scripts/bench/run_multi_gpu_probe.py:107`).
One process per GPU; tracks are dealt longest-processing-time-first; there is NO data-path
collective — the only communication is batch completion: a barrier and one `all_gather_object` of
per-track summaries (a few hundred bytes each; over RCCL/xGMI on the GPU box, over gloo in the CPU
tests).
"""
from __future__ import annotations

import hashlib
from typing import Dict, List, Optional, Sequence

import numpy as np


def assign_tracks(durations_s: Sequence[float], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first: returns, per rank, the indices of its tracks (stable for ties)."""
    world_size = max(1, int(world_size))
    order = sorted(range(len(durations_s)), key=lambda i: (-float(durations_s[i]), i))
    loads = [0.0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += float(durations_s[i])
    for r in range(world_size):
        out[r].sort()
    return out


def summarize(track_index: int, sample_boundaries: Sequence[int], seconds: float, timings: Optional[Dict[str, float]] = None) -> Dict:
    b = np.asarray(list(sample_boundaries), dtype=np.int64)
    return {"track": int(track_index), "n_boundaries": int(b.size), "boundaries_sha1": hashlib.sha1(b.tobytes()).hexdigest(),
            "audio_seconds": float(seconds), "timings": dict(timings or {})}


def gather_summaries(local: List[Dict], group=None) -> List[Dict]:
    """Batch completion: barrier + all_gather_object; every rank returns the full, track-ordered list."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return sorted(local, key=lambda d: d["track"])
    dist.barrier(group=group)
    buckets: List[Optional[List[Dict]]] = [None] * dist.get_world_size(group)
    dist.all_gather_object(buckets, local, group=group)
    merged = [d for b in buckets for d in (b or [])]
    return sorted(merged, key=lambda d: d["track"])


class TrackPipeline:
    """Software pipeline over the tracks of ONE GPU: `depth` worker threads, each with its own high-priority HIP stream and its own
    `SeamlessSplitter` (they share the read-only backend / U-Net weights), plus ONE U-Net stream for all of them.  A worker queues its
    track's separation on the U-Net stream under `separation_gate` (held only while queueing), then runs the track's host-bound tail
    (VAD bookkeeping, pause detection, guard, boundary policy: ~20 ms of small kernels and synchronisation round trips) on its own
    stream while the next track's U-Net - already waiting in the queue - runs: the GPU never idles between two tracks, and two U-Nets
    never run at once (that only slows both: the package power is the shared budget).  Tracks stay independent: no state is shared
    between workers, results come back in submission order and are bit-identical to one-at-a-time processing."""

    def __init__(self, splitters: Sequence, device) -> None:
        import threading
        import torch
        self._device = torch.device(device)
        self._workers = [(sp, torch.cuda.Stream(device=self._device, priority=-1)) for sp in splitters]
        self.separation_gate = threading.Lock()      # one U-Net on the GPU at a time: pass it to split_track(separation_gate=...)
        # Optional: ONE stream for every worker's separation (split_track(unet_stream=...)).  The stream orders the U-Nets of
        # consecutive tracks itself, so the gate is held only while a track's launches are queued and the next U-Net waits in the
        # queue behind the running one; the workers' own streams (tails, VAD, detection: small kernels with host round trips) have
        # the higher priority.
        self.unet_stream = torch.cuda.Stream(device=self._device, priority=0)

    @property
    def depth(self) -> int:
        return len(self._workers)

    def run(self, jobs: Sequence) -> List:
        """`jobs[i](splitter)` -> result; every job runs under one worker's stream; returns the results in job order."""
        import queue
        import threading
        import torch
        todo: "queue.Queue" = queue.Queue()
        for i, job in enumerate(jobs):
            todo.put((i, job))
        results: List = [None] * len(jobs)
        errors: List[BaseException] = []

        def work(splitter, stream) -> None:
            with torch.cuda.device(self._device), torch.cuda.stream(stream):
                while not errors:
                    try:
                        i, job = todo.get_nowait()
                    except queue.Empty:
                        break
                    try:
                        results[i] = job(splitter)
                    except BaseException as exc:      # surfaced to the caller below; the other worker stops at its next job
                        errors.append(exc)
                        break
                stream.synchronize()

        threads = [threading.Thread(target=work, args=w, name=f"audiocut-track-{k}") for k, w in enumerate(self._workers)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results


__all__ = ["assign_tracks", "summarize", "gather_summaries", "TrackPipeline"]
