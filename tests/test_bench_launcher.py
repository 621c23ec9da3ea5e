"""bench.py's N-rank path without a launcher around it: `python bench.py --gpus N` must become N ranks.

CPU tier: `--dry-ranks` makes every rank join the process group (gloo here), take part in one all-reduce and rank 0 print
what it saw — the same self-spawn, rendezvous and rank bookkeeping the GPU run goes through, with nothing rendered."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE json line, from rank 0
    return json.loads(lines[0]), r.stderr


def test_gpus_flag_spawns_that_many_ranks():
    j, _ = _run(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--dry-ranks"])
    assert j == {"n_gpus": 2, "ranks_seen": 2, "backend": "gloo", "reduce": "gloo (host)"}


def test_native_reduce_route_is_one_process_of_n_contexts():
    """`--native-reduce`: the one-process route — Scene::build(devices) + mi_film_reduce — one command away from
    an 8-GPU lease; the dry run prints its plan: N contexts, distinct devices -> RCCL, a shared device -> the rank-ordered add"""
    j, _ = _run(["--gpus", "4", "--native-reduce", "--dry-ranks"])
    assert j["n_gpus"] == 4 and j["ranks_seen"] == 4 and j["devices"] == [0, 1, 2, 3] and "RCCL" in j["reduce"] and "one process" in j["route"]
    j, _ = _run(["--gpus", "3", "--native-reduce", "--share-gpu", "--dry-ranks"])
    assert j["devices"] == [0, 0, 0] and "device add" in j["reduce"]
    j, _ = _run(["--gpus", "1", "--native-reduce", "--dry-ranks"])
    assert j["ranks_seen"] == 1 and "none" in j["reduce"]


def test_one_gpu_is_one_process():
    j, _ = _run(["--gpus", "1", "--dry-ranks"])
    assert j["n_gpus"] == 1 and j["ranks_seen"] == 1 and j["reduce"] is None


def test_under_a_launcher_the_launchers_rank_count_runs():
    """the driver's form: torch.distributed.run around bench.py --gpus N — no second spawn, n_gpus = WORLD_SIZE"""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.spawn_command(2, ["--gpus", "2", "--backend", "gloo", "--dry-ranks"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "127.0.0.1" in cmd and cmd.count("--gpus") == 1
    e = dict(os.environ); e.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_reduce_is_labelled_by_the_backend_in_use():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.reduce_label("nccl") == "RCCL" and "gloo" in bench.reduce_label("gloo")


def test_bench_names_the_film_kernel_that_runs():
    """VERDICT r04 (measurement hygiene 12): the line's kernel_ms named k_film_groups while k_film_columns<4,2> ran. The label now comes
    from the library: mi_counters::film_kernel says which block replay mi_render launched (a GPU test checks the counter itself:
    test_gpu_parity.py::test_film_replay_kernels_agree)."""
    import bench
    assert bench.film_kernel_name(24, 0) == "k_film_blocks" and bench.film_kernel_name(24, 4) == "k_film_blocks"
    assert [bench.film_kernel_name(16, k) for k in (1, 2, 3, 4)] == ["k_film_groups", "k_film_columns", "k_film_quads", "k_film_lanes"]


def test_a_film_that_is_not_the_oracles_makes_the_parity_field_false():
    """the bench line's `parity.match` must turn false when a single bit of the timed film moves. CPU tier: the hashing /
    comparison itself against the committed digests (the GPU run of the default line is what makes it true: profiles/r06_bench_default_line.json)."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import bench
    for which in ("c2", "c3", "c4", "c5"):
        path, key = bench.GOLDEN_FRAMES[which]
        rec = json.load(open(os.path.join(ROOT, path)))[key]
        film = torch.zeros(1080 * 1920 * 5, dtype=torch.float32)
        out = bench.film_parity(film, which, rec["samples"], rec["segments"])
        assert out["golden_sha256"] == rec["sha256"] and out["match"] is False and out["counts_match"] is True and "error" not in out
        out = bench.film_parity(film, which, rec["samples"] + 1, rec["segments"])         # the film AND the counts must be the oracle's
        assert out["match"] is False and out["counts_match"] is False
    # the comparison is the digest of the float32 bytes: a buffer that hashes to the golden value matches, one flipped bit does not
    import hashlib
    film = torch.arange(64, dtype=torch.float32)
    bench.GOLDEN_FRAMES["_t"] = ("tests/golden/round3.json", "c2_full_1920x1080_512spp")
    try:
        rec = json.load(open(os.path.join(ROOT, "tests/golden/round3.json")))["c2_full_1920x1080_512spp"]
        real = bench.film_parity(film, "_t")
        assert real["match"] is False and real["film_sha256"] == hashlib.sha256(film.numpy().tobytes()).hexdigest()
        flipped = film.clone(); flipped.view(torch.int32)[7] ^= 1
        assert bench.film_parity(flipped, "_t")["film_sha256"] != real["film_sha256"]
    finally:
        del bench.GOLDEN_FRAMES["_t"]
