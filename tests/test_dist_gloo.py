"""Multi-process layout of McSAS.analyse on CPU: world_size-2 gloo run of the rep sharding and the
single end-of-run all-gather (mcsas_amd/dist.py).  The per-rep payload is synthetic here (no GPU);
what is checked is that every rank ends up with all reps in rep order."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mcsas_amd.dist import shard_reps, gather_results


def test_shard_reps_partitions_in_order():
    for R, W in ((50, 8), (7, 2), (3, 4), (400, 8), (1, 1)):
        blocks = [shard_reps(R, W, r) for r in range(W)]
        assert sum(c for _, c in blocks) == R
        pos = 0
        for first, count in blocks:
            assert first == pos
            pos += count
        assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


def _fake_rep(r, N, P, Q):
    rs = np.random.RandomState(1000 + r)
    return dict(contribs=rs.rand(N, P), fit=rs.rand(Q), chisq=np.array([float(r) + 0.5]))


def _worker(rank, world, port, R, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, P, Q = 6, 2, 5
    first, count = shard_reps(R, world, rank)
    reps = [_fake_rep(first + i, N, P, Q) for i in range(count)]
    local = {k: np.stack([x[k] for x in reps]) if count else np.zeros((0,) + _fake_rep(0, N, P, Q)[k].shape)
             for k in ("contribs", "fit", "chisq")}
    full = gather_results(local, R)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **full)
    dist.destroy_process_group()


@pytest.mark.parametrize("R", [5, 4, 1])
def test_two_rank_gather_restores_rep_order(tmp_path, R):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, R, str(tmp_path)), nprocs=2, join=True)
    want = [_fake_rep(r, 6, 2, 5) for r in range(R)]
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        for k in ("contribs", "fit", "chisq"):
            np.testing.assert_array_equal(got[k], np.stack([w[k] for w in want]))


def test_single_process_is_a_passthrough():
    out = gather_results(dict(a=np.arange(6.0).reshape(3, 2)), 3)
    np.testing.assert_array_equal(out["a"], np.arange(6.0).reshape(3, 2))


def _run_bench(tmp_path, gpus, extra, tag, env_extra=None):
    """bench.py's own multi-rank code path on CPU: MCSAS_BENCH_DRY=1 swaps the GPU plan for a payload that
    depends only on (seed, global repetition index); gloo carries the all-gather.  --gpus N with no launcher
    around makes bench.py start its N ranks itself."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = os.path.join(str(tmp_path), tag + ".npz")
    env = dict(os.environ, MCSAS_BENCH_DRY="1", MCSAS_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1",
           "--dump", dump] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), np.load(dump)


@pytest.mark.parametrize("extra", [["--scaling", "strong"], ["--scaling", "strong", "--config", "5"]])
def test_bench_multi_rank_path_equals_single_rank(tmp_path, extra):
    """Strong scaling shards the config's repetitions (50, or config 5's 100) over the ranks with chain id =
    global repetition index: the gathered result of 2 ranks equals the 1-rank run repetition for repetition,
    and the JSON line reports the ranks it actually saw."""
    one, a1 = _run_bench(tmp_path, 1, extra, "one")
    two, a2 = _run_bench(tmp_path, 2, extra, "two")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["ranks_seen"] == 2 and two["scaling"] == "strong"
    assert two["config"]["reps_total"] == one["config"]["reps_total"]
    assert two["value"] is None and two["data"] == "dry-run"          # never a measurement
    for k in ("contribs", "fit", "chisq", "scaling", "background"):
        np.testing.assert_array_equal(a1[k], a2[k])


def test_bench_forced_collective_at_world_size_one(tmp_path):
    """MCSAS_BENCH_FORCE_DIST=1: one rank, but through the process group, the packed all-gather (timed: gather_ms) and the
    all-reduce of the timings — the switch the GPU suite uses to run the RCCL path on a one-GPU box."""
    one, a1 = _run_bench(tmp_path, 1, ["--scaling", "strong"], "plain")
    os.environ["MCSAS_BENCH_FORCE_DIST"] = "1"
    try:
        forced, a2 = _run_bench(tmp_path, 1, ["--scaling", "strong"], "forced")
    finally:
        del os.environ["MCSAS_BENCH_FORCE_DIST"]
    assert "gather_ms" not in one and forced["gather_ms"]["ranks"] == 1 and forced["gather_ms"]["backend"] == "gloo"
    assert forced["config"]["ranks_seen"] == 1 and forced["n_gpus"] == 1
    for k in ("contribs", "fit", "chisq", "scaling", "background"):
        np.testing.assert_array_equal(a1[k], a2[k])


def test_bench_default_scaling_follows_the_config(tmp_path):
    """Config 2 is named per GPU (50 repetitions on 1 MI355X): weak by default; configs 3-5 are named as totals over 8 GPUs: strong."""
    two, _ = _run_bench(tmp_path, 2, [], "c2")
    assert two["scaling"] == "weak" and two["config"]["reps_total"] == 100 and two["config"]["reps_per_gpu"] == 50
    five, _ = _run_bench(tmp_path, 2, ["--config", "5"], "c5")
    assert five["scaling"] == "strong" and five["config"]["reps_total"] == 100 and five["config"]["reps_per_gpu"] == 50


def test_bench_runs_the_named_totals_of_configs_3_to_5_over_the_ranks(tmp_path):
    """`bench.py --gpus N` with the default --config 2 also runs BASELINE's configs 3-5 as they are named — 200 / 400 / 100
    repetitions IN ALL, sharded over the ranks (mcsas.py:214 is the loop being sharded) — and reports them per config with the ranks
    it saw; the N = 1 line carries the same totals on one rank (the denominator of a 1 -> N ratio).  Dry run: no measurement."""
    one, _ = _run_bench(tmp_path, 1, [], "t1", {"MCSAS_BENCH_DRY_CONFIGS": "1"})
    two, _ = _run_bench(tmp_path, 2, [], "t2", {"MCSAS_BENCH_DRY_CONFIGS": "1"})
    for k, total in (("3", 200), ("4", 400), ("5", 100)):
        a, b = one["configs"][k], two["configs"][k]
        assert a["reps_total"] == b["reps_total"] == total and a["scaling"] == b["scaling"] == "strong"
        assert a["ranks_seen"] == 1 and a["reps_rank0"] == total
        assert b["ranks_seen"] == 2 and b["reps_rank0"] == total // 2 and b["n_gpus"] == 2
        assert a["value"] is None and b["value"] is None               # dry run: never a measurement


def test_bench_refuses_a_world_size_it_was_not_asked_for(tmp_path):
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MCSAS_BENCH_DRY="1", MCSAS_BENCH_BACKEND="gloo", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout
