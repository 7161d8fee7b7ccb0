# experiment: does the ORDER of the queue (registrations sorted by the position of their initial guess) change the stream's rate?
import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import numpy as np
from elimaloc_amd import synth
from elimaloc_amd.registration import Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod, Scan
from concurrent.futures import ThreadPoolExecutor
ctx = Context(0)
world = synth.make_world(10_000_000, seed=1001)
vm = VoxelHashMap(1.0, 30, ctx); vm.AddPoints(world)
N = 2048
def gen(i):
    sc, Tt = synth.make_scan(world, 131072, seed=2002 + i, max_range=60.0, noise=0.01)
    return sc, synth.perturb(Tt, seed=3003 + i, max_trans=0.15, max_rot_deg=0.5)
synth.make_scan(world, 16, seed=1)
with ThreadPoolExecutor(16) as p: g = list(p.map(gen, range(N)))
scans = [Scan(ctx, a[0]) for a in g]; T0s = [a[1] for a in g]
reg = Registration(RegistrationConfig(icp_method=IcpMethod.P2P), ctx)
def rate(order, tag):
    pk = reg.pack_inputs([scans[i] for i in order], [T0s[i] for i in order])
    for _ in range(2): reg.RunRegisterStream(pk[0], vm, pk[1], slots=256, raw=True)
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(8): reg.RunRegisterStream(pk[0], vm, pk[1], slots=256, raw=True)
    ctx.synchronize(); el = time.perf_counter() - t
    print(f"{tag:10s} {N * 8 / el:10.1f} registrations/s", flush=True)
ident = list(range(N))
xy = np.array([T[:2, 3] for T in T0s])
cell = np.floor((xy - xy.min(0)) / 40.0).astype(int)
snake = np.lexsort((np.where(cell[:, 0] % 2 == 0, cell[:, 1], -cell[:, 1]), cell[:, 0]))  # boustrophedon over 40 m cells
for rep in range(2):
    rate(ident, "random")
    rate(list(snake), "sorted")
