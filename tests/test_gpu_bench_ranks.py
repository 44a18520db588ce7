"""bench.py with more than one rank, as a child process with a time limit: two gloo ranks sharing the one GPU on a
ragged batch (shards of different size once chose different numbers of ring buffers, and the fences between ring
segments -- collectives -- then waited for each other for ever), and the RCCL all-reduce + graph-replay path with
a process group of one rank.  Reference analogue of the sharded step: in-process batch slicing,
reference rnnt/keras_model.py:203-206."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra=None, timeout=280):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(env_extra or {}))
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_two_gloo_ranks_on_a_ragged_batch_finish_and_cover_the_batch():
    line = _bench(["--gpus", "2", "--backend", "gloo", "--config", "5", "--batch", "12", "--steps", "3", "--warmup", "1",
                   "--no-cpu-baseline", "--no-legs"], {"MASTER_PORT": "29581"})
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "strong"
    ranks = line["per_rank"]
    assert len(ranks) == 2 and sum(r["utterances"] for r in ranks) == 12
    assert sum(r["cells"] for r in ranks) == line["config"]["total_cells"]
    # shards of different size: the agreed ring length is one number for both ranks
    assert ranks[0]["padded_rows"] != ranks[1]["padded_rows"]
    assert line["config"]["ring_buffers"] >= 1
    assert line["value"] > 0 and line["ms_per_step"] > 0


def test_rccl_all_reduce_and_graph_replay_with_a_group_of_one():
    line = _bench(["--gpus", "1", "--backend", "nccl", "--config", "3", "--batch", "8", "--steps", "4", "--warmup", "1",
                   "--no-cpu-baseline", "--no-legs"], {"BENCH_FORCE_PROCESS_GROUP": "1", "MASTER_PORT": "29583"})
    assert line["n_gpus"] == 1 and line["backend"] == "nccl"
    assert line["config"]["hip_graph"] is True
    assert line["config"]["total_cells"] == 8 * 400 * 101
    assert 0.0 < line["roofline"]["frac"] < 1.0
