"""Host logic of bench.py that needs no GPU: starting the rank processes,
refusing a mismatched launch, the byte accounting of the roofline figures."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_spawner_refuses_to_run_after_gpu_init(monkeypatch):
    import torch
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    with pytest.raises(RuntimeError, match="already initialised"):
        bench.spawn_ranks(2, ["--gpus", "2"])


def test_bare_multi_gpu_launch_starts_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` with no launcher starts two rank processes itself; without a GPU each of them
    stops with an error, and the parent must say so and exit non-zero instead of printing a one-GPU line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("checks the no-GPU failure path")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr and "exited with status" in r.stderr
    assert r.stdout.strip() == ""


def test_world_size_must_match_gpus(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit, match="WORLD_SIZE=4"):
        bench.main(["--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE")
    with pytest.raises(SystemExit, match="--gpus must be"):
        bench.main(["--gpus", "0"])


def test_kernel_shares_add_up_to_the_bytes_per_cell():
    for V in (62, 512, 1024, 4096):
        assert sum(bench.kernel_bytes_per_cell(V).values()) == bench.bytes_per_cell(V) == 8 * V + 16


def test_global_lengths_do_not_depend_on_the_rank():
    cfg = bench.CONFIGS[5]
    a = bench.make_lengths(cfg, 0xC0FFEE + 5)
    b = bench.make_lengths(cfg, 0xC0FFEE + 5)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()
    assert a[0].min() >= 50 and a[0].max() <= 1000 and a[1].min() >= 10 and a[1].max() <= 200


def test_counter_bytes_are_quoted_only_for_the_kernels_they_were_measured_on(tmp_path):
    import json
    from rnnt_amd import build
    path = tmp_path / "traffic.json"
    ent = {"B": 64, "prune_log2": -28.0, "csrc_sha16": build.sources_digest(), "step_hbm_bytes": 1.0}
    path.write_text(json.dumps({"config3": ent}))
    assert bench.counter_bytes_for(3, 64, -28.0, str(path)) == ent
    assert bench.counter_bytes_for(3, 32, -28.0, str(path)) == {}
    assert bench.counter_bytes_for(3, 64, -24.0, str(path)) == {}
    path.write_text(json.dumps({"config3": dict(ent, csrc_sha16="0" * 16)}))   # profile of an older build
    assert bench.counter_bytes_for(3, 64, -28.0, str(path)) == {}
    path.write_text(json.dumps({"config3": {k: v for k, v in ent.items() if k != "csrc_sha16"}}))  # unstamped
    assert bench.counter_bytes_for(3, 64, -28.0, str(path)) == {}
