"""CPU dry run of bench.py's multi-GPU control flow (round-4 verdict, item 3): `bench.main()` itself -- argument handling,
the three legs, the side measurements, the watchdog, the failure path, the one JSON line -- at world 8 over gloo, with the
device side replaced by tests/shard_fakes.FakeRuntime (the product's ShardedSae over the oracle's kernels).  No number in the
line is a measurement; what is pinned is that the first real 8-GPU run cannot die in Python and what its line looks like
(DESIGN.md section 6 documents the schema)."""
import json
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, out, argv, fault):
    for p in (REPO, REPO / "tests", REPO / "multimodal-sae_amd"):
        sys.path.insert(0, str(p))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    if fault.get("watchdog_s"):
        os.environ["MSAE_BENCH_WATCHDOG_S"] = str(fault["watchdog_s"])
    import bench
    import shard_fakes

    rt = shard_fakes.FakeRuntime(stall_rank=fault.get("stall_rank"), raise_rank=fault.get("raise_rank"))
    bench.main(argv + ["--json-out", out], rt=rt)


def _run(tmp_path, world, argv, fault=None):
    out = str(tmp_path / "line.json")
    try:
        mp.spawn(_rank, args=(world, _free_port(), out, argv, fault or {}), nprocs=world, join=True)
    except Exception as e:  # a rank ended with a non-zero code (the injected-failure case): the line must exist anyway
        if not fault:
            raise
        assert os.path.exists(out), f"no JSON line after {e}"
    return json.loads(open(out).read())


def test_bench_world_8_control_flow_and_schema(tmp_path):
    res = _run(tmp_path, 8, ["--gpus", "8", "--steps", "2", "--warmup", "1", "--tokens", "80", "--k", "8", "--batches", "2"])
    assert "error" not in res, res.get("error")
    assert res["n_gpus"] == 8 and res["rccl_world"] == 8 and res["collective_backend"] == "gloo" and "dry_run" in res
    assert res["scaling"] == "strong" and res["headline"].startswith("feature-sharded: ")
    assert res["value"] > 0 and res["ms_per_step"] > 0 and res["higher_is_better"] is True
    assert res["sharded_bit_identical_to_single_gpu_on_256_tokens"] is True
    assert res["replicas"]["scaling"] == "weak" and res["replicas"]["tokens_per_step"] == 8 * 80
    modes = res["shard_modes"]
    for name, colls in (("per_shard_topk", {"all_gather_pairs", "all_gather_reconstruction"}),
                        ("candidate_exchange", {"all_to_all_records", "all_gather_results", "all_gather_reconstruction"})):
        m = modes[name]
        assert m["bit_identical_256"] is True and m["ms_per_step"] > 0 and m["ms_per_step_no_recon_gather"] > 0
        assert colls <= set(m["collective_ms"]), m["collective_ms"]
    assert modes["per_shard_topk"]["k_loc"] >= 1 and modes["candidate_exchange"]["candidates_per_shard"] >= 8
    for key in ("metric", "unit", "steps", "warmup", "vs_baseline", "dtype", "data", "config", "dither"):
        assert key in res
    assert "feature-sharded x8" in res["config"]["parallelism"] and res["config"]["tokens_per_step"] == 80


def test_bench_world_4_one_token_fewer_than_ranks(tmp_path):
    """T = 3 < G = 4: a rank without a token in the sharded decode and in the candidate exchange."""
    res = _run(tmp_path, 4, ["--gpus", "4", "--steps", "1", "--warmup", "1", "--tokens", "3", "--k", "8", "--batches", "1"])
    assert "error" not in res, res.get("error")
    assert res["shard_modes"]["candidate_exchange"]["bit_identical_256"] is True


@pytest.mark.parametrize("fault", [{"raise_rank": 2, "watchdog_s": 60}, {"stall_rank": 3, "watchdog_s": 15}])
def test_bench_line_survives_a_failing_or_wedged_leg(tmp_path, fault):
    """The second feature-sharded leg raises on rank 2, or wedges on rank 3 (its peers park in a collective): rank 0 still
    prints ONE line -- the headline of the last completed leg, the error named -- through the store monitor (seconds) or the
    watchdog."""
    res = _run(tmp_path, 4, ["--gpus", "4", "--steps", "1", "--warmup", "1", "--tokens", "20", "--k", "8", "--batches", "1"],
               fault=fault)
    assert "error" in res and res["value"] > 0
    assert res["headline"] == "feature-sharded: per_shard_topk"          # the leg that completed before the faulty one
    assert res["shard_modes"]["per_shard_topk"]["bit_identical_256"] is True


def test_numa_pinned_cpu_baseline_leg_runs_and_reports():
    """bench.py's cpu_baseline also times the reference port pinned to one NUMA node (oracle/cpu_baseline_worker.py, a process of
    its own so that the affinity mask precedes torch's thread pool); here: a tiny shape, the JSON it must hand back."""
    import bench

    rec = bench.numa_pinned_baseline(256, 2048, 8, 32, 2)
    assert "error" not in rec, rec
    if "skipped" in rec:
        return
    assert rec["tokens_per_s"] > 0 and rec["threads"] >= 1 and rec["affinity"] >= rec["threads"] and rec["node"] == 0
