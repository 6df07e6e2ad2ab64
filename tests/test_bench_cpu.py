"""bench.py's launcher logic without a device: `--gpus N` starts N ranks itself when no launcher did, and never reports fewer devices as N."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DMVIO_BENCH_SHARE_DEVICE")}


def test_more_gpus_than_devices_is_refused():
    import torch
    n = torch.cuda.device_count() + 2
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=200, env=_env())
    assert p.returncode == 2 and not p.stdout.strip()
    assert b"--gpus %d asked for" % n in p.stderr and b"refusing" in p.stderr


def test_launcher_world_size_must_match_gpus():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=200, env=dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert p.returncode != 0 and not p.stdout.strip() and b"WORLD_SIZE=2" in p.stderr
