"""bench.py as the driver launches it: the JSON line is the LAST thing on stdout and the only JSON line, also under
`python -m torch.distributed.run` with the RCCL route forced on a single rank (the same code path every rank of an 8-GPU run
takes: process group on nccl = RCCL, communicator through the C ABI, one ncclBroadcast of the packed weight slab)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _json_lines(stdout):
    out = []
    for line in stdout.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def _env(**extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra)
    return env


def test_bench_under_torch_distributed_run_with_the_rccl_route():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29571", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-e2e"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(DEMON_FORCE_DIST="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    recs = _json_lines(r.stdout)
    assert len(recs) == 1, r.stdout[-2000:]
    assert json.loads(lines[-1]) == recs[0]                       # the JSON line is the last thing the job prints
    rec = recs[0]
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["unit"] == "pairs/s" and rec["scaling"] == "weak"
    assert rec["config"]["rccl_nranks"] == 1 and rec["config"]["weights_broadcast_route"] == "rccl"
    assert rec["outputs_finite"] is True and rec["value"] > 100
    assert abs(rec["value"] - 32 * 2 / (rec["ms_per_step"] * 2e-3)) < 1e-6 * rec["value"]
    # every rank's own clock and lane calibration travel in the line (what makes the first 8-GPU run diagnosable)
    pr = rec["per_rank"]
    assert len(pr["ranks"]) == 1 and pr["ranks"][0]["rank"] == 0 and pr["lanes"] == [rec["config"]["lanes"]]
    assert pr["pairs_per_s_min"] == pr["pairs_per_s_max"] >= rec["value"] * (1 - 1e-6)           # the reported time is the MAX over ranks
    m = pr["ranks"][0]["lanes_mapping"]
    assert m["lanes"] == rec["config"]["lanes"] and 0 <= m["placeholder_streams"] <= 5
    assert pr["ranks"][0]["lanes_calibration_pairs_per_s"] == rec["config"]["lanes_calibration_pairs_per_s"]
    assert rec["value_single_lane"] == rec["single_lane"]["pairs_per_s"] > 100


def test_bench_default_line_has_roofline_and_host_to_host_rates():
    """the line the driver records (cpu_baseline left out here: it is 30 s of host work and has its own protocol)"""
    r = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    recs = _json_lines(r.stdout)
    assert len(recs) == 1
    rec = recs[0]
    assert "configs[2]" in rec["config"]["workload"] and rec["dtype"] == "f32" and rec["vs_baseline"] is None
    for key in ("roofline", "roofline_worst", "roofline_family"):
        e = rec[key]
        assert e["bound"] == "mfma" and e["peak"] == 157.3 and e["unit"] == "TFLOP/s"
        assert abs(e["frac"] - e["achieved"] / e["peak"]) < 1e-9
        assert 0.0 < e["frac"] <= 1.0, (key, e["frac"])                                  # executed multiply-adds: a roofline fraction
        assert e["algorithmic_frac"] >= e["frac"] - 1e-12
        assert abs(e["frac"] - e["algorithmic_frac"] * e["mfma_flops_executed_per_algorithmic"]) < 1e-9
        if rec["config"]["lanes"] > 1:   # the entry's own figures are those of the regime `value` ran in; one launch at a time under stand_alone
            f = e["in_flight"]
            assert e["regime"].startswith("in flight") and f["lanes"] == rec["config"]["lanes"] and f["frac"] == e["frac"]
            a = e["stand_alone"]
            assert 0.0 < a["frac"] <= 1.0 and a["algorithmic_frac"] >= a["frac"] - 1e-12 and a["avg_launch_ms"] > 0
        else:
            assert e["regime"].startswith("stand-alone")
    assert 0.0 < rec["pipeline_mfma_executed_frac"] <= rec["pipeline_mfma_frac"] <= 1.0
    assert abs(rec["pipeline_mfma_frac"] - rec["value"] * 30.353e9 / 157.3e12) < 1e-9
    assert 1 <= rec["config"]["lanes"] <= 5 and rec["config"]["steps_in_flight_per_gpu"] == rec["config"]["lanes"]
    one = rec["single_lane"]
    assert one["lane0_outputs_match"] is True and 0.5 * rec["value"] < one["pairs_per_s"] < 1.02 * rec["value"]
    # round 6: the same lanes with the image-only layers of the iterative nets evaluated once per pass -- reported BESIDE the headline, with the
    # outputs compared bit for bit in the run; the headline's metric name and flops stay those of the full graph
    h = rec["image_features_hoisted"]
    assert h["outputs_bit_identical"] is True and rec["value_image_features_hoisted"] == h["pairs_per_s"] > 0.9 * rec["value"]
    assert "hoisted" not in rec["metric"] and h["note"].startswith("NOT the headline")
    m = rec["config"]["lanes_mapping"]
    if rec["config"]["lanes"] > 1:   # the calibration left the lanes on a mapping it measured (plateau rule) or reproduced its winner
        assert m["reproduced"] is True and 1 <= m["attempts"] <= 17, m
    x = rec["extra"]
    assert x["end_to_end_pairs_per_s"] > 0
    p = x["pipelined"]
    assert p["pinned"] is True and p["outputs_equal_resident_run"] is True and 2 <= p["contexts"] <= 5
    assert p["pairs_per_s"] > x["end_to_end_pairs_per_s"]                                  # overlap beats synchronous pageable copies
    assert 0.5 < p["frac_of_resident"] < 1.5      # (two batches in flight hide per-launch latencies: the pipelined rate may exceed the one-stream rate)


def test_headline_protocol_parity_under_torch_distributed_run():
    """tools/lane_parity.py as a rank of the driver's launch (process group on nccl = RCCL first, so its streams shift the stream ->
    hardware-queue mapping, and the package asks for 16 hardware queues): every pair of every kept lane against the CPU oracle"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29573", os.path.join("tools", "lane_parity.py"), "--lanes", "5", "--dist"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    rec = _json_lines(r.stdout)[-1]
    assert rec["dist"] is True and rec["hw_queues"]["requested"] == "16" and rec["hw_queues"]["runtime_was_up"] is False
    assert 1 <= rec["lanes_kept"] <= 5 and rec["mapping"]["verified_pairs_per_s"] > 0
    assert rec["outputs_finite"] and rec["second_round_bit_equal"]
    assert rec["pairs_checked"] == 32 * rec["lanes_kept"] and rec["worst_rel_l1"] < 1e-3, rec["worst_at"]


def test_two_processes_calibrate_their_lanes_concurrently_on_one_gpu():
    """the closest reachable stand-in for eight ranks calibrating at once: two processes on this one GPU, each running the whole
    LaneGroup.calibrate sweep at the same time (batch 8 so that both fit comfortably); both must finish, verify their winner and
    report it, and both must still compute finite, stable outputs"""
    cmd = [sys.executable, os.path.join("tools", "lane_parity.py"), "--lanes", "4", "--batch", "8", "--no-oracle"]
    procs = [subprocess.Popen(cmd, cwd=ROOT, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=1500) for p in procs]
    recs = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, (so[-2000:], se[-3000:])
        recs.append(_json_lines(so)[-1])
    assert recs[0]["pid"] != recs[1]["pid"]
    for rec in recs:
        assert 1 <= rec["lanes_kept"] <= 4 and rec["mapping"]["verified_pairs_per_s"] > 0 and rec["mapping"]["attempts"] >= 1
        assert rec["calibration_cells"] >= 4 and rec["outputs_finite"] and rec["second_round_bit_equal"]
