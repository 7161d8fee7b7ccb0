"""bench.py's N > 1 entry: `python bench.py --gpus N` must itself launch N ranks (one per GPU), refuse a box with fewer GPUs,
and hand every rank exactly its contiguous shard of every scan while each rank generates only 1/N of the scans.

The GPU-free part of that path (--dry-launch: launcher -> torch.distributed.run -> gloo rendezvous -> sharded generation +
all-to-all) runs here with two processes; the device part is `bench.py`'s normal run on the GPU box."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def expected(n_batch, npts, map_points, world_size):
    """inputs.sha1 and, per rank, (points, sha1) of its shards in registration order -- generated here, in one process."""
    sys.path.insert(0, ROOT)
    from elimaloc_amd import synth
    from elimaloc_amd import dist as elm_dist
    world = synth.make_world(map_points, seed=1001)
    digests = []
    shard = [hashlib.sha1() for _ in range(world_size)]
    pts = [0] * world_size
    for i in range(n_batch):
        sc, Tt = synth.make_scan(world, npts, seed=2002 + i, max_range=60.0, noise=0.01)
        T0 = synth.perturb(Tt, seed=3003 + i, max_trans=0.15, max_rot_deg=0.5)
        h = hashlib.sha1(sc.tobytes())
        h.update(np.ascontiguousarray(T0).tobytes())
        digests.append(h.digest())
        # W > 1: the ranks hold contiguous parts of the spatially ORDERED scan (locality-aware sharding, round 6)
        parts = elm_dist.spatial_shards(sc, world_size) if world_size > 1 else [sc]
        assert sum(len(q) for q in parts) == npts and sorted(map(tuple, np.concatenate(parts))) == sorted(map(tuple, sc))  # a partition of the scan
        for r in range(world_size):
            lo, hi = npts * r // world_size, npts * (r + 1) // world_size
            assert len(parts[r]) == hi - lo
            shard[r].update(parts[r].tobytes())
            pts[r] += hi - lo
    return hashlib.sha1(b"".join(digests)).hexdigest(), [(pts[r], shard[r].hexdigest()) for r in range(world_size)]


def test_gpus_2_launches_two_ranks_with_their_shards():
    npts, batch, mp = 1001, 3, 30000  # an odd scan size: ragged shard bounds
    p = run(["--gpus", "2", "--dry-launch", "--batch", str(batch), "--scan-points", str(npts), "--map-points", str(mp)])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # exactly one JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["dry_launch"] is True and line["value"] is None
    assert line["n_gpus"] == 2
    assert [r["rank"] for r in line["ranks"]] == [0, 1]
    assert [r["local_rank"] for r in line["ranks"]] == [0, 1]
    assert len({r["pid"] for r in line["ranks"]}) == 2  # two processes
    sha, shards = expected(batch * 2, npts, mp, 2)
    assert line["inputs"]["registrations"] == batch * 2
    assert line["inputs"]["sha1"] == sha  # the sharded generation is the one-process generation, bit for bit
    for r in range(2):
        assert (line["ranks"][r]["shard_points"], line["ranks"][r]["shard_sha1"]) == shards[r]


def test_one_rank_dry_run_has_the_same_inputs():
    p = run(["--gpus", "1", "--dry-launch", "--batch", "6", "--scan-points", "1001", "--map-points", "30000"])
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    sha, shards = expected(6, 1001, 30000, 1)
    assert line["n_gpus"] == 1 and line["inputs"]["sha1"] == sha
    assert (line["ranks"][0]["shard_points"], line["ranks"][0]["shard_sha1"]) == shards[0]


def test_gpus_2_without_two_gpus_fails_loudly():
    # (the device count without importing torch into the test process: its bundled RCCL beside the one the product library dlopens
    # ends the interpreter with a double free at exit)
    import ctypes
    try:
        n = ctypes.c_int(0)
        if ctypes.CDLL("libamdhip64.so").hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value >= 2:
            pytest.skip("this box has two GPUs")
    except OSError:
        pass
    p = run(["--gpus", "2", "--steps", "1", "--warmup", "0"], timeout=300)
    assert p.returncode != 0
    assert "refusing to run" in p.stderr and "--gpus 2" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]  # no bench line at all


def test_gpus_must_equal_world_size():
    # a rank environment that disagrees with --gpus (e.g. torch.distributed.run --nproc-per-node 1 bench.py --gpus 2)
    p = run(["--gpus", "2", "--dry-launch"], env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_PORT": "29999",
                                                        "MASTER_ADDR": "127.0.0.1"}, timeout=300)
    assert p.returncode != 0 and "!= WORLD_SIZE" in p.stderr
    # and the plain one-process form cannot claim more GPUs than processes either
    p = run(["--gpus", "1", "--dry-launch", "--batch", "1", "--scan-points", "64", "--map-points", "30000"],
            env_extra={"WORLD_SIZE": "2"}, timeout=300)
    assert p.returncode != 0 and "!= WORLD_SIZE" in p.stderr


@pytest.mark.gpu
def test_n2_flow_rehearsed_on_one_gpu():
    """The WHOLE `bench.py --gpus 2` flow on a one-GPU box (ELM_BENCH_SHARED_GPU: both ranks on device 0, sums exchanged over gloo instead
    of RCCL, which refuses two ranks on one device): launcher, sharded input path (Hilbert-contiguous shards through the all-to-all),
    sharded registrations with one exchange per iteration, hard-guess leg, replica leg, the single-process device-group child, the line.
    It measures nothing (`rehearsal: true`); it shows that the pieces the driver's first multi-GPU run executes do execute, and that the
    three process models agree on the poses."""
    p = run(["--gpus", "2", "--batch", "24", "--slots", "8", "--steps", "1", "--warmup", "1", "--scan-points", "6001", "--map-points", "200000"],
            env_extra={"ELM_BENCH_SHARED_GPU": "1"}, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rehearsal"] is True and line["value"] > 0 and line["config"]["registrations_per_step"] == 48
    assert line["replica"]["max_abs_pose_diff_vs_sharded"] < 1e-9             # whole registrations per rank vs the sharded ones
    sp = line["single_process"]
    assert "error" not in sp and sp["devices"] == 2 and sp["iterations_match"] is True and sp["max_abs_pose_diff_vs_ranks"] < 1e-9
    assert len(lines[0]) < 6000
