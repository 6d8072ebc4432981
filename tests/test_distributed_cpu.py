"""world_size-2 gloo rehearsal of the N>1 path: track sharding + batch-completion barrier/all_gather_object."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_dir: str) -> None:
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from audio_cut_amd import batch
    durations = [240.0, 240.0, 1800.0, 60.0, 240.0]
    mine = batch.assign_tracks(durations, world)[rank]
    local = []
    for t in mine:       # stand-in for SeamlessSplitter.split_track on this rank's GPU
        rng = np.random.default_rng(100 + t)
        bounds = np.concatenate(([0], np.sort(rng.integers(1, int(durations[t] * 44100), 7)), [int(durations[t] * 44100)]))
        local.append(batch.summarize(t, bounds.tolist(), durations[t], {"step_s": 0.1 * (t + 1)}))
    merged = batch.gather_summaries(local)
    assert [d["track"] for d in merged] == list(range(len(durations)))
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array([d["boundaries_sha1"] for d in merged]))
    dist.destroy_process_group()


def test_two_rank_track_sharding(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a = np.load(tmp_path / "rank0.npy"); b = np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b) and len(a) == 5          # every rank ends with the same complete, ordered summary


def test_c3_single_gpu_table_agrees_with_the_cpu_oracle(golden_dir=None):
    """tests/golden/c3_n1_sha1.json (the per-track result of the 32 C3 tracks on one MI355X, what `bench.py --gpus N` checks
    every rank's tracks against) hashes the SAME integers the CPU oracle produced for the two tracks it was run on
    (c3_seed100_oracle.npz, c3_seed131_oracle.npz: 4 min of oracle time each): guard boundaries and manifest cuts."""
    import hashlib
    import json
    from pathlib import Path
    import numpy as np
    gd = Path(__file__).resolve().parent / "golden"
    table = json.loads((gd / "c3_n1_sha1.json").read_text())["tracks"]
    assert sorted(int(k) for k in table) == [2] + list(range(100, 132))      # the C2 bench track (seed 2) + the 32 C3 tracks
    for seed in (100, 131):
        g = np.load(gd / f"c3_seed{seed}_oracle.npz")
        assert int(g["seed"]) == seed and float(g["seconds"]) == 240.0
        assert hashlib.sha1(g["sample_boundaries"].astype(np.int64).tobytes()).hexdigest() == table[str(seed)]["boundaries_sha1"]
        assert hashlib.sha1(g["cuts"].astype(np.int64).tobytes()).hexdigest() == table[str(seed)]["cuts_sha1"]
        assert len(g["sample_boundaries"]) == table[str(seed)]["n_boundaries"]
