"""ncclAllReduce with MORE THAN ONE rank: two processes, one real 2-rank RCCL communicator.

The build/test box has a single MI355X, so both ranks sit on GPU 0.  RCCL refuses two ranks of one communicator on the same
device ("Duplicate GPU detected") -- when it does, the test is an expected failure that RECORDS the refusal text, so the run log
shows why the N > 1 collective could not execute here; on a box where the communicator forms (>= 2 GPUs are not required by
this test, only a build that tolerates duplicates) the sharded stream registration must agree with the unsharded one.
The data path of two real ranks (shards, reduce-only / solve-only launches, identical slot refills) is covered without RCCL by
tests/test_gpu_parity.py::test_two_ranks_on_one_gpu, the protocol on CPU by tests/test_distributed.py.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_process_rccl_communicator(tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_rccl_rank.py"), str(r), "2", str(tmp_path)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate()[0] + "\n[timeout]")
    codes = [p.returncode for p in procs]
    if any(c == 3 for c in codes):
        msgs = [open(tmp_path / f"rank{r}.err").read() for r in range(2) if (tmp_path / f"rank{r}.err").exists()]
        dup = [ln for o in outs for ln in o.splitlines() if "Duplicate GPU" in ln]
        pytest.xfail("RCCL refused a 2-rank communicator on one device: " + " | ".join(msgs + dup[:1]))
    assert codes == [0, 0], "\n".join(outs)
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["it"], b["it"])  # both ranks solved the same all-reduced sums
