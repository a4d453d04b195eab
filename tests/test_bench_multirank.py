"""bench.py --gpus N: the launcher starts N rank processes itself; the config-3 workload (`--workload ont`) cuts ONE ragged
batch into contiguous shards balanced by cumulative bases and every rank runs the HIP path on its shard.  Rehearsed here
on one GPU (two ranks share device 0, gloo for the count all-reduce): the ranks' outputs, concatenated in rank order,
must equal the unsharded run (the reference's worker pool, src/main.rs:57,65-79, has the same property per read)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, tmp, tag, workload="ont"):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
           "--no-other-mode", "--verify-reads", "50", "--dump-shard", os.path.join(tmp, tag)] + extra
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [x for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_equal_unsharded(tmp_path):
    tmp = str(tmp_path)
    one = run_bench(["--gpus", "1", "--reads", "6000"], tmp, "one")
    two = run_bench(["--gpus", "2", "--reads", "3000", "--single-device", "--backend", "gloo"], tmp, "two")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["counts"] == {**two["counts"], "xor_hash_rank0": one["counts"]["xor_hash_rank0"]}  # whole-job totals agree
    assert two["config"]["reads_total"] == 6000 and 1.0 <= two["config"]["largest_shard_over_mean"] < 1.01
    assert one["verified_vs_oracle"] and two["verified_vs_oracle"]
    # the JSON proves how many ranks the process group saw, and the one real exchange step (count: all-to-all by hash prefix) ran
    # (a leg that did not run leaves no key in the line)
    assert one.get("collective") is None and two["collective"]["world_size_seen"] == 2 and two["collective"]["ranks"] == [0, 1]
    assert two["per_rank"]["wall_ms_per_step_max"] >= two["per_rank"]["wall_ms_per_step_min"] > 0
    assert one.get("downstream_count") is None and two["downstream_count"]["n_keys"] == two["counts"]["kminmers"]
    assert two["downstream_count"]["exchange"].startswith("all_to_all_single")
    assert one["scaling"] == "weak" and one.get("end_to_end") is None  # (ont workload: the PCIe legs belong to the c2 line)
    a = np.load(os.path.join(tmp, "one.rank0.npz"))
    parts = [np.load(os.path.join(tmp, "two.rank%d.npz" % r)) for r in range(2)]
    assert int(parts[0]["first_base"]) == 0 and int(parts[1]["first_base"]) == int(parts[0]["n_bases"])  # contiguous shards of one stream
    for f in ("hash", "start", "end", "rev"):
        assert (np.concatenate([p[f] for p in parts]) == a[f]).all(), f
    km = np.concatenate([parts[0]["km_off"][:-1], parts[1]["km_off"] + parts[0]["km_off"][-1]])
    assert (km == a["km_off"]).all()


@pytest.mark.gpu
def test_hifi_workload_two_ranks_equal_one(tmp_path):
    """BASELINE configs[3] under the multi-GPU launcher: rank r generates and processes the HiFi-like reads r * n .. (r + 1) * n - 1 of the
    one numbered sequence of reads (weak scaling), so two ranks of 2000 reads together produce exactly what one rank of 4000 does."""
    tmp = str(tmp_path)
    one = run_bench(["--gpus", "1", "--reads", "4000"], tmp, "h1", workload="hifi")
    two = run_bench(["--gpus", "2", "--reads", "2000", "--single-device", "--backend", "gloo"], tmp, "h2", workload="hifi")
    assert one["config"]["name"] == two["config"]["name"] == "BASELINE configs[3]"
    assert one["verified_vs_oracle"]["ok"] and two["verified_vs_oracle"]["ok"]
    assert {k: one["counts"][k] for k in ("bases", "minimizers", "kminmers")} == {k: two["counts"][k] for k in ("bases", "minimizers", "kminmers")}
    a = np.load(os.path.join(tmp, "h1.rank0.npz"))
    parts = [np.load(os.path.join(tmp, "h2.rank%d.npz" % r)) for r in range(2)]
    for f in ("hash", "start", "end", "rev"):
        assert (np.concatenate([p[f] for p in parts]) == a[f]).all(), f
    km = np.concatenate([parts[0]["km_off"][:-1], parts[1]["km_off"] + parts[0]["km_off"][-1]])
    assert (km == a["km_off"]).all()


@pytest.mark.gpu
def test_strong_scaling_cuts_one_fixed_batch(tmp_path):
    tmp = str(tmp_path)
    one = run_bench(["--gpus", "1", "--scaling", "strong", "--total-reads", "6000"], tmp, "s1")
    two = run_bench(["--gpus", "2", "--scaling", "strong", "--total-reads", "6000", "--single-device", "--backend", "gloo"], tmp, "s2")
    assert one["scaling"] == two["scaling"] == "strong"
    assert one["config"]["reads_total"] == two["config"]["reads_total"] == 6000 and one["counts"]["bases"] == two["counts"]["bases"]
    assert one["counts"]["kminmers"] == two["counts"]["kminmers"]


@pytest.mark.gpu
def test_rccl_that_cannot_come_up_does_not_cost_the_line(tmp_path):
    """default backend (RCCL for tensors in HBM, gloo for host memory) with two ranks on ONE device: RCCL refuses ("Duplicate GPU"),
    every rank agrees to move the control words (barrier, max of the times, count sums) to gloo, and the line says which it was --
    the extraction itself has no collective"""
    tmp = str(tmp_path)
    one = run_bench(["--gpus", "1", "--reads", "4000"], tmp, "f1")
    two = run_bench(["--gpus", "2", "--reads", "2000", "--single-device"], tmp, "f2")
    assert two["collective"]["backend"].startswith("gloo (RCCL unusable") and two["collective"]["world_size_seen"] == 2
    assert two["verified_vs_oracle"]["ok"] and {k: one["counts"][k] for k in ("bases", "kminmers")} == {k: two["counts"][k] for k in ("bases", "kminmers")}
    assert two["downstream_count"]["n_keys"] == two["counts"]["kminmers"]


def test_more_ranks_than_gpus_fails_loudly():
    """no GPU (or fewer than --gpus) visible: every rank says so and exits non-zero before anything touches a device"""
    import torch

    n = torch.cuda.device_count()
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 2)], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "GPU(s) visible" in (p.stderr + p.stdout)


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
