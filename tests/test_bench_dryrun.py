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
        # (r05) what makes the first real N-GPU run diagnosable from its one line
        m = d["multi_gpu"]
        assert len(m["per_rank_ms_per_step"]) == gpus and len(m["per_rank_k1_avg_launch_ms"]) == gpus
        assert m["rank_skew_ms_per_step"] >= 0.0 and m["status_allreduce_latency_us"] > 0.0
        assert "gloo" in m["status_allreduce_backend"]
        w = d["weak_extra"]
        assert w["global_batch"] == batch * gpus and w["batch_per_gpu"] == batch
    else:
        assert d["weak_extra"] is None


def test_panel_launch_periods_split_overlapping_intervals():
    """bench.py's `roofline.avg_launch_ms` is the COMPLETION PERIOD of the panel launches (r05: the resident launches of the
    two batch groups overlap on their own streams): e1_i - max(e0_i, e1_{i-1}) in completion order — the plain e1 - e0
    for disjoint launches, the union of the busy intervals split at the completions otherwise."""
    import bench

    class Ev:                                   # stand-in for a HIP event: a time stamp in ms
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t
    mk = lambda s, e: (Ev(s), Ev(e), 6, 32)
    # disjoint launches with gaps: own durations
    per, raw, nb = bench._k1_periods([mk(0.0, 5.0), mk(6.0, 11.5), mk(12.0, 17.0)], 6)
    assert nb == 32 and [round(x * 1e3, 6) for x in per] == [5.0, 5.5, 5.0] and per == raw
    # overlapping launches (enqueued while the previous one runs): periods sum to the union, own intervals do not
    # (launches start and complete in the order they were enqueued: a resident launch holds the machine until its queue
    #  is empty; events arrive in any order in the list)
    per, raw, _ = bench._k1_periods([mk(2.0, 10.5), mk(0.0, 5.0), mk(7.0, 16.0), mk(17.0, 18.0)], 6)
    assert [round(x * 1e3, 6) for x in per] == [5.0, 5.5, 5.5, 1.0]                 # completion order: 5.0, 10.5, 16.0, 18.0
    assert abs(sum(per) - 17.0e-3) < 1e-12 and sum(raw) > sum(per)                  # union = [0, 16] + [17, 18]
    # other panel widths are not mixed in
    per, _, _ = bench._k1_periods([mk(0.0, 5.0), (Ev(1.0), Ev(2.0), 16, 8)], 6)
    assert len(per) == 1
    assert bench._pct([3.0, 1.0, 2.0], 0.5) == 2.0 and bench._pct([], 0.5) is None
