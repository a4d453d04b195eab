"""CPU tier: pieces of bench.py that need no GPU -- the id that ties `roofline.traffic` to the kernel sources it was measured on,
and the workloads the command line offers."""
import json
import os
import shutil
import sys

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_kernel_source_id_follows_the_device_sources(tmp_path, monkeypatch):
    a = bench.kernel_source_id()
    assert a == bench.kernel_source_id() and len(a) == 16
    # the same function over a copy of the tree in which one kernel source differs by one byte
    dst = tmp_path / "rust-seq2kminmers_amd" / "csrc"
    dst.mkdir(parents=True)
    src = os.path.join(ROOT, "rust-seq2kminmers_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".h")) or f == "Makefile":
            shutil.copy(os.path.join(src, f), dst / f)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.kernel_source_id() == a  # objects, libraries and listings do not count
    with open(dst / "s2k_tile_impl.h", "a") as f:
        f.write("\n")
    assert bench.kernel_source_id() != a


def test_committed_traffic_file_names_its_sources():
    tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert tj["mode"] == "hpc" and tj["n_bases"] == 10_000_000_000
    assert isinstance(tj.get("kernel_source_id"), str) and len(tj["kernel_source_id"]) == 16  # bench.py compares it with the tree's
    assert 12e9 < tj["hbm_bytes_per_step"] < 30e9


def test_workloads_on_the_command_line(monkeypatch):
    for wl in ("c2", "ont", "hifi"):
        monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", wl, "--no-other-configs"])
        a = bench.parse_args()
        assert a.workload == wl and a.gpus == 1 and a.no_other_configs
