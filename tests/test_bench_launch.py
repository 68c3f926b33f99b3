"""`python bench.py --gpus N` (N > 1) outside a torch.distributed job starts the job itself: the
parent never imports torch or touches HIP, launches `python -m torch.distributed.run` with N
workers on 127.0.0.1, relays rank 0's JSON line and exits with the job's status.  Here, without a
GPU, the workers fail loudly (no CPU fallback) and the parent must report exactly that; the GPU
box runs the same launch for real (tests/test_gpu_bench_contract.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parent_launches_the_job_and_relays_its_status():
    import torch

    if torch.cuda.is_available():  # the GPU box: covered by the contract tests
        import pytest

        pytest.skip("CPU-only check")
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
                          "1", "--warmup", "0", "--nodes", "1000", "--backend", "gloo",
                          "--share-device"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode != 0
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]  # no line was invented
    # both workers were started by torch.distributed.run and died of the missing GPU
    assert "No AMD GPU" in res.stderr or "no HIP device" in res.stderr.lower(), res.stderr[-1500:]
    assert "bench.py FAILED" in res.stderr


def test_watchdog_ends_a_hung_job_and_leaves_a_record():
    """A worker that never gets past a phase (GN2V_BENCH_STALL: the stand-in for an RCCL bootstrap
    that hangs): after --job-timeout the parent terminates the job's whole process group, prints
    ONE JSON line -- error, the phase each rank had reached, the job's last stderr lines -- and
    exits non-zero; no worker survives."""
    import json
    import time

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GN2V_BENCH_STALL"] = "start:600"   # before anything touches a GPU: runs here as well
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
                          "1", "--warmup", "0", "--nodes", "1000", "--backend", "gloo",
                          "--share-device", "--job-timeout", "20"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert time.time() - t0 < 120
    assert res.returncode == 124, (res.returncode, res.stderr[-1500:])
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["error"] == "timeout" and rec["n_gpus"] == 2
    assert rec["phase_reached_per_rank"] == {"0": "start", "1": "start"}, rec
    assert "value" not in rec  # no number was invented
    # the workers are gone (the stall would have kept them for ten minutes)
    out = subprocess.run(["ps", "-eo", "args"], capture_output=True, text=True).stdout
    assert not [l for l in out.splitlines() if "bench.py" in l and "--job-timeout 20" in l
                and "pytest" not in l], out


def test_parent_does_not_import_torch():
    """The launcher path runs before `import torch`: a process that never initialises the GPU."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("launch_job(args)") < main.index("import torch")
    launcher = src[src.index("def launch_job(args):"):src.index("def main():")]
    assert "import torch" not in launcher and "torch.distributed.run" in launcher
