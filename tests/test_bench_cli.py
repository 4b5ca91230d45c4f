"""bench.py's command line without a GPU: a --gpus / WORLD_SIZE mismatch is a clear non-zero exit."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_mismatch_is_a_clear_error():
  env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
                       env=env, capture_output=True, text=True, timeout=300)
  assert out.returncode == 2
  assert "torch.distributed.run" in out.stderr and "--gpus 4" in out.stderr
  assert not out.stdout.strip()
  env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = "2", "0", "0"
  out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
  assert out.returncode == 2 and "WORLD_SIZE=2" in out.stderr
