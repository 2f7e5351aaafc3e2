"""REAL multi-GPU runs: one process per physical GPU, the library's RCCL transport between different devices
(SURVEY.md section 8(e) correctness test: the unitig set from n in {2, 4, 8} GPUs == the oracle's).  Skips on a box with
one GPU (the driver's test box); `python -m pytest tests/test_gpu_multi.py -m gpu` on any multi-GPU MI355X node proves the
transport.  The single-device coverage of the same code (N contexts on one GPU over a loop-back transport, 1-rank RCCL)
is tests/test_gpu_parity.py; the gloo coverage on CPU is tests/test_dist_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _n_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _worlds():
    n = _n_gpus()
    return [w for w in (2, 4, 8) if w <= n]


@pytest.mark.parametrize("k,reads,read_len,cfg,replicated", [
    (31, 400000, 150, 3, 0), (31, 400000, 150, 3, 1), (55, 150000, 150, 4, 0), (55, 150000, 150, 4, 1), (127, 8000, 1000, 5, 0)])
def test_one_process_per_gpu_rccl(tmp_path, k, reads, read_len, cfg, replicated):
    worlds = _worlds()
    if not worlds:
        pytest.skip("needs >= 2 GPUs: %d visible (the RCCL transport between two devices cannot run here)" % _n_gpus())
    for world in worlds:
        out = tmp_path / ("w%d.json" % world)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "CDBG_FORCE_MULTI"):
            env.pop(v, None)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(29600 + world + (k % 7) * 10 + replicated), os.path.join(ROOT, "tests", "mgpu_worker.py"),
               "--k", str(k), "--reads", str(reads), "--read-len", str(read_len), "--cfg", str(cfg), "--replicated", str(replicated), "--out", str(out)]
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        r = json.load(open(out))
        assert r["world"] == world and r["set_equal"] and r["distinct_equal"] and r["solid_equal"] and r["verify_equal"], r
        assert all(b > 0 for b in r["comm_bytes"]), r            # bytes really moved between the devices


def test_worker_script_with_one_rank(tmp_path):
    """the same launcher + worker with ONE rank (CDBG_FORCE_MULTI: the multi-rank code path through a 1-rank RCCL
    communicator), so that the script the multi-GPU test depends on is itself exercised on every box"""
    if _n_gpus() < 1:
        pytest.skip("no GPU")
    out = tmp_path / "w1.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CDBG_FORCE_MULTI="1")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29655",
           os.path.join(ROOT, "tests", "mgpu_worker.py"), "--k", "31", "--reads", "100000", "--out", str(out)]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    r = json.load(open(out))
    assert r["world"] == 1 and r["set_equal"] and r["distinct_equal"] and r["solid_equal"] and r["verify_equal"], r
