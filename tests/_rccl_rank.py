"""One rank of tests/test_rccl_two_ranks.py: python tests/_rccl_rank.py <rank> <nranks> <dir>.

Creates a context on GPU 0, joins an `nranks`-rank RCCL communicator (the id travels through <dir>/id.bin), registers its
contiguous shard of a few seeded scans in stream mode and writes the poses to <dir>/rank<r>.npz.  Exit code 0 = ran,
3 = ncclCommInitRank refused (the message is written to <dir>/rank<r>.err).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elimaloc_amd import synth  # noqa: E402
from elimaloc_amd._lib import ElmError  # noqa: E402
from elimaloc_amd.dist import shard_bounds  # noqa: E402
from elimaloc_amd.registration import Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod, Scan  # noqa: E402

rank, nranks, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ctx = Context(0)
idf = os.path.join(d, "id.bin")
if rank == 0:
    open(idf + ".tmp", "wb").write(Context.comm_unique_id())
    os.replace(idf + ".tmp", idf)
t0 = time.time()
while not os.path.exists(idf):
    if time.time() - t0 > 120:
        sys.exit(4)
    time.sleep(0.05)
try:
    ctx.comm_init(rank, nranks, open(idf, "rb").read())
except ElmError as e:
    open(os.path.join(d, f"rank{rank}.err"), "w").write(str(e))
    sys.exit(3)
world = synth.make_world(100000, seed=1001)
vm = VoxelHashMap(1.0, 30, ctx)
vm.AddPoints(world)
scans, T0s = [], []
for i in range(5):
    sc, Tt = synth.make_scan(world, 4000 + 900 * i, seed=900 + i)
    lo, hi = shard_bounds(len(sc), rank, nranks)
    scans.append(Scan(ctx, sc[lo:hi], n_total=len(sc)))
    T0s.append(synth.perturb(Tt, seed=950 + i, max_trans=0.05 + 0.05 * i, max_rot_deg=0.3 + 0.2 * i))
out = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), ctx).RunRegisterStream(scans, vm, T0s, slots=2)
np.savez(os.path.join(d, f"rank{rank}.npz"), T=np.array([r["T"] for r in out]), it=np.array([r["iterations"] for r in out]))
ctx.comm_destroy()
ctx.close()
