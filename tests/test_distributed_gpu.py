"""Batch completion over RCCL on the GPU box (the reference's only multi-device tool is
scripts/bench/run_multi_gpu_probe.py:107-139).  The box has ONE GPU and RCCL refuses two ranks on one device, so this runs
bench.py's distributed branch with world_size = 1 on the `nccl` backend in a fresh child process: librccl loads,
init_process_group(device_id=...), barrier, the MAX all-reduce, all_gather_object over a device, destroy_process_group and a clean exit
are then known good on this image before the first multi-GPU run needs them.  The N > 1 control flow itself is covered on the CPU
(tests/test_distributed_cpu.py, gloo, world 2)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_bench_distributed_branch_on_rccl_with_one_rank():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               AC_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--track-seconds", "40",
                        "--cpu-baseline-seconds", "0"], env=env, capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["batch_completion"]["backend"] == "nccl"
    assert line["batch_completion"]["summaries_gathered"] == 1 and line["tracks_completed"] == 1


@pytest.mark.gpu
def test_gather_summaries_over_rccl_with_one_rank():
    code = (
        "import os, torch, torch.distributed as dist\n"
        "from audio_cut_amd import batch\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group(backend='nccl', device_id=torch.device('cuda', 0))\n"
        "dist.barrier()\n"
        "local = [batch.summarize(3, [0, 40, 80], 1.0), batch.summarize(1, [5], 2.0)]\n"
        "out = batch.gather_summaries(local)\n"
        "assert [d['track'] for d in out] == [1, 3] and out[1]['n_boundaries'] == 3, out\n"
        "t = torch.tensor([1.5], dtype=torch.float64, device='cuda:0'); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t) == 1.5\n"
        "dist.destroy_process_group()\n"
        "print('rccl ok')\n")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=str(ROOT))
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert p.returncode == 0 and "rccl ok" in p.stdout, (p.stdout[-500:], p.stderr[-2000:])
