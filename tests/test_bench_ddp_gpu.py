"""bench.py's multi-rank control flow on a one-GPU box: 2 ranks share cuda:0 and exchange
gradients over gloo (test hook X3D_BENCH_SINGLE_DEVICE / X3D_BENCH_BACKEND); the production
launch is one rank per GPU over RCCL (backend "nccl"), which needs a multi-GPU node."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_on_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, X3D_BENCH_SINGLE_DEVICE="1", X3D_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--frames", "4", "--size", "64",
           "--no-cpu-baseline"]      # kernel timing stays ON: its eager steps contain collectives every rank must join
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 4
    assert d["value"] > 0 and d["unit"] == "clips/s"
    # the data-parallel section the multi-GPU scaling record explains itself with
    dp = d["data_parallel"]
    assert dp["rccl_ranks"] == 2 and dp["backend"] == "gloo" and len(dp["buckets_bytes"]) == 2
    assert len(dp["allreduce_ms_per_bucket_standalone"]) == 2 and all(t > 0 for t in dp["allreduce_ms_per_bucket_standalone"])
    assert sum(dp["buckets_bytes"]) == 4 * 3794322        # the flat fp32 gradient of X3D-M (15.18 MB)
