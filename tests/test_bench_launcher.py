"""`python bench.py --gpus N` must be able to start its own N ranks: the parent spawns one fresh
process per rank under torch.distributed.run before touching any GPU, relays rank 0's single JSON line and returns
the children's exit code.  Rehearsed here without hardware (ANNP_BENCH_DRYRUN=1: ranks on CPU over gloo, the whole
step loop -- halo exchange, re-homing, thermo all-reduce -- except the force evaluation, which has no CPU path)."""
import json
import os
import subprocess
import sys

from annp_testlib import ROOT


def run_bench(*argv, timeout=600):
    env = dict(os.environ, ANNP_BENCH_DRYRUN="1", OMP_NUM_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_two_ranks_started_by_bench_itself():
    r = run_bench("--gpus", "2", "--steps", "4", "--warmup", "1", "--cells", "12", "--rebuild-every", "2", "--thermo", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["measured_n"] == [2] and out["steps"] == 4 and out["warmup"] == 1
    cfg = out["config"]
    assert cfg["world_size"] == 2 and len(cfg["atoms_rank"]) == 2 and sum(cfg["atoms_rank"]) == cfg["atoms"] == 2 * 12 ** 3
    assert all(g > 0 for g in cfg["ghosts_rank"]) and cfg["halo_bytes_per_step"] > 0
    assert out["value"] is None and "rehearsal" in out          # nothing is measured without a GPU
    assert out["metric"].startswith("atom-steps/sec (whole node), bcc-Fe ANNP")


def test_exit_code_of_the_ranks_is_returned():
    """4 cells along x cut in two: slabs thinner than the halo -> every rank raises -> the launcher must not report success"""
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "4")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_single_rank_needs_no_launcher():
    r = run_bench("--gpus", "1", "--steps", "2", "--warmup", "0", "--cells", "6", "--rebuild-every", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 1 and out["config"]["atoms_rank"] == [432]


def test_eight_ranks_the_driver_configuration():
    """The driver's last configuration, N = 8, rehearsed here on CPU: eight ranks over gloo through
    launch_ranks, slabs 3 cells = 8.57 A thick (the thinnest a one-neighbour halo allows), re-planning and thermo all-reduces
    included; rank 0's line carries what tells a slow rank from a slow wire."""
    r = run_bench("--gpus", "8", "--steps", "3", "--warmup", "1", "--cells", "24", "--rebuild-every", "2", "--thermo", "2", timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 8 and cfg["world_size"] == 8 and out["scaling"] == "strong"
    assert len(cfg["atoms_rank"]) == 8 and sum(cfg["atoms_rank"]) == cfg["atoms"] == 2 * 24 ** 3
    assert all(g > 0 for g in cfg["ghosts_rank"])
    pr = out["per_rank"]
    assert len(pr["step_ms"]) == 8 and all(v > 0 for v in pr["step_ms"])
    assert len(pr["halo_ms"]["forward"]) == 8 and len(pr["halo_ms"]["reverse"]) == 8 and all(v >= 0 for v in pr["halo_ms"]["forward"])
    assert all(len(v) == 8 for v in pr["kernel_ms"].values())
    assert out["mini_md"] is not None


def test_world_size_mismatch_fails_loudly():
    """--gpus must be the size of the job: a rank started with another WORLD_SIZE exits non-zero with one line"""
    env = dict(os.environ, ANNP_BENCH_DRYRUN="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--cells", "8"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()


def test_a_wire_that_hangs_ends_the_run_by_itself():
    """A rank that never takes part in the first exchange (stand-in for a hung RCCL wire): every rank gives up after
    ANNP_BENCH_WIRE_TIMEOUT with one line on stderr and a non-zero exit code, instead of sitting until somebody kills the job."""
    import time
    env = dict(os.environ, ANNP_BENCH_DRYRUN="1", ANNP_BENCH_WIRE_TIMEOUT="8", ANNP_BENCH_TEST_HANG_RANK="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cells", "8", "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "0", "--secondary", "0"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode != 0
    assert "did not complete within" in r.stderr
    assert time.time() - t0 < 120
