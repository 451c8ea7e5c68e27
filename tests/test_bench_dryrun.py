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
    for attempt in range(2):
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=140, cwd=ROOT)
        # (gloo's own teardown aborts once in a long while on a loaded box — "terminate called without an active
        #  exception" after the line was printed; one retry, any other failure is reported)
        if proc.returncode == 0 or b"terminate called without an active exception" not in proc.stderr:
            break
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


def test_flat_scalars_repeat_the_nested_blocks_inside_roofline():
    """ the driver's record keeps the SCALARS of `roofline` only: the SURVEY 8d-conformant full-matrix
    figure, the bare stream, the launch percentiles and one scalar set per secondary BASELINE config are repeated there."""
    import bench
    out = {"roofline": {"frac": 0.77, "survey_8d": {"frac": 0.8, "achieved": 6400.0, "avg_launch_ms": 10.7,
                                                    "eigpairs_per_s": 990.0, "ms_per_step": 388.0, "traffic": 7.0e10,
                                                    "algorithmic_bytes_per_launch": 6.9e10},
                        "stream_read": {"GBps": 6800.0}, "launch_ms_p10_p50_p90": [5.2, 5.6, 5.9],
                        "standalone_whole_batch_launch": {"avg_launch_ms": 10.9, "frac": 0.79}},
           "configs": {"c0": {"ms_per_step": 80.0, "value": 75.0, "exacteig": {"ms_per_step": 5.2},
                              "roofline": {"frac": 2e-4}, "check": {"ok": True},
                              "cpu_baseline": {"value": 10.0, "exacteig_value": 370.0}},
                       "c3": {"error": "boom"},
                       "c5w": {"ms_per_step": 95.0, "value": 2700.0, "check": {"ok": True},
                               "roofline": {"frac": 0.6, "avg_launch_ms": 3.5, "frac_of_fp32_matrix_peak": 0.5,
                                            "traffic": None, "algorithmic_bytes_per_launch": 1.72e10},
                               "cpu_baseline": {"value": 57.0, "full_config_seconds_extrapolated": 430.0}}}}
    bench._flat_scalars(out)
    r = out["roofline"]
    assert r["frac"] == 0.77 and r["frac_survey_8d"] == 0.8 and r["survey_8d_frac"] == 0.8
    assert r["survey_8d_ms"] == 10.7 and r["survey_8d_eigpairs_per_s"] == 990.0 and r["stream_read_GBps"] == 6800.0
    assert (r["p10_ms"], r["p50_ms"], r["p90_ms"]) == (5.2, 5.6, 5.9)
    assert r["c0_ms"] == 80.0 and r["c0_exacteig_ms"] == 5.2 and r["c0_cpu"] == 10.0 and r["c0_cpu_exacteig"] == 370.0
    assert "c3_ms" not in r
    assert r["c5w_ms"] == 95.0 and r["c5w_frac"] == 0.6 and r["c5w_launch_ms"] == 3.5 and r["c5w_mfma_frac"] == 0.5
    assert r["c5w_cpu"] == 57.0 and r["c5w_cpu_full_config_s"] == 430.0 and r["c5w_check_ok"] == 1
    assert "frac_survey_8d" in r["note"]
    for v in r.values():                               # nothing new that the driver would drop
        assert isinstance(v, (int, float, str, dict, list)) or v is None
    bench._flat_scalars({"roofline": None})            # dry runs carry no roofline
