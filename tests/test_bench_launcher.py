"""bench.py's own multi-rank launcher, as far as it can be checked without a GPU: `python bench.py --gpus N` (N > 1, no launcher
around it) must start N ranks as child processes, relay their output and return their worst exit code -- and must itself never
touch the GPU.  The run that actually measures is `-m gpu`: tests/test_gpu_multirank.py::test_bench_py_gpus_2_launches_its_own_ranks."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLEAN = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


@pytest.mark.skipif(torch.cuda.is_available(), reason="on a GPU box this command is the real benchmark (run by the -m gpu test)")
def test_gpus_2_starts_two_ranks_and_returns_their_exit_code():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--envs-per-gpu", "256", "--pmc", "off", "--cpu-baseline", "0"], cwd=ROOT, env=CLEAN, capture_output=True, text=True,
                       timeout=300)
    # no GPU here: both ranks fail when they first touch it, the launcher reports that and fails too (it did not raise SystemExit
    # about WORLD_SIZE, and it did not die on a GPU call of its own)
    assert r.returncode != 0
    assert "rank exit codes [1, 1]" in r.stderr, r.stderr[-1500:]
    assert "No HIP GPUs are available" in r.stderr and "launch with torch.distributed.run" not in r.stderr
    assert r.stdout.strip() == ""


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box with fewer GPUs than ranks")
def test_gpus_n_over_rccl_needs_n_gpus():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], cwd=ROOT, env=CLEAN, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 2 and "needs 8 GPUs" in r.stderr and r.stdout.strip() == ""


def test_a_wrong_world_size_is_reported():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--pmc", "off"], cwd=ROOT,
                       env=dict(CLEAN, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_traffic_from_counters_prices_scattered_and_coalesced_reads():
    """bench.py's HBM traffic from FETCH_SIZE / WRITE_SIZE (KB): gfx950 counts a coalesced stream's reads at half and scattered 64-byte
    records exactly (profiles/r04_hbm_scatter_calibration.txt), so only the coalesced share is doubled."""
    sys.path.insert(0, ROOT)
    import bench
    c = {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 3000.0}
    assert bench.traffic_bytes(None) is None and bench.traffic_bytes({"FETCH_SIZE": 1.0}) is None
    assert bench.traffic_bytes(c) == (2 * 1000 + 3000) * 1024                        # every read a coalesced stream
    assert bench.traffic_bytes(c, 0) == (1000 + 3000) * 1024                         # every read a scattered record
    assert bench.traffic_bytes(c, 200 * 1024) == (1000 + 100 + 3000) * 1024          # + half the coalesced streams' bytes
    assert bench.traffic_bytes(c, 10 ** 12) == bench.traffic_bytes(c)                # never more than all reads doubled
