"""not-gpu: `bench.py --gpus N --dry-run` — the launcher / process-group / timing / JSON plumbing of the multi-GPU bench
on CPU ranks over gloo (make the N > 1 path fail-safe before a node ever sees it).  No performance
number is produced or claimed."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("gpus,batch", [(2, 5), (1, 3)])
def test_bench_dry_run_prints_one_json_line(gpus, batch):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-run", "--steps", "2", "--warmup", "1",
           "--batch", str(batch)]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=280, cwd=ROOT)
    assert proc.returncode == 0, proc.stderr.decode()[-2000:]
    lines = [ln for ln in proc.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                          # rank 0 prints exactly one line, nobody else anything
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["value"] is None and d["roofline"] is None
    assert d["n_gpus"] == gpus and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == batch and d["config"]["comm_world_size"] == gpus
    assert d["config"]["batch_per_gpu"] == -(-batch // gpus)          # rank 0 holds the larger shard
    assert d["check"]["ok"]
    if gpus > 1:
        assert d["config"]["comm_backend"] == "gloo"
        w = d["weak_extra"]
        assert w["global_batch"] == batch * gpus and w["batch_per_gpu"] == batch
    else:
        assert d["weak_extra"] is None
