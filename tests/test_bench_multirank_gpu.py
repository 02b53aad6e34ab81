"""The N > 1 code path of bench.py as the driver will launch it, on a box with ONE GPU: two ranks under torch.distributed.run
share the device (F3DG_DIST_BACKEND=gloo: frames are gathered through host memory; the measured configuration is nccl = RCCL).
Checks the JSON line rank 0 prints: the first real `--gpus 8` must not die on a code path no test ran."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun_bench(extra, timeout=900):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, F3DG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 prints ONE JSON line, the other rank none
    return json.loads(lines[0])


def test_bench_c2_two_ranks_gloo(gpu_device):
    d = _torchrun_bench(["--gaussians", "65536", "--views", "24", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "views/s" and d["value"] > 0 and d["higher_is_better"] is True
    assert abs(d["value"] - 2 * 24 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]       # whole-job views / max-over-ranks time
    assert d["config"]["views"] == 24 and "RCCL gather" in d["config"]["parallelism"]
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["kernel"].startswith("render3s_fwd_kernel<")      # the name the LIBRARY reports for what it launched
    assert d["ranks_seen"] == 2 and d["dist_backend"] == "gloo"
    assert "cpu_baseline" not in d and "with_d2h" not in d


def test_bench_c4_two_ranks_gloo(gpu_device):
    d = _torchrun_bench(["--workload", "c4", "--images", "2", "--res", "64", "--backbone", "bf16"])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["images_per_gpu"] == 2
    assert abs(d["value"] - 2 * 2 * 8 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
