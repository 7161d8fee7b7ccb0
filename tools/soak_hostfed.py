"""Developer soak (GPU): the host-fed stream (three HIP streams, per-group events, bounded upload look-ahead) against the resident stream of
the same scans -- many repetitions, several slot counts, pageable and page-locked sources, ragged sizes: every result bit-identical."""
import sys, numpy as np
sys.path.insert(0, '.')
from elimaloc_amd import synth
from elimaloc_amd.registration import Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod, Scan, PinnedBuffer
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = Context(0)
world = synth.make_world(2_000_000, seed=1001)
rng = np.random.default_rng(5)
for method in (0, 2):
    m = IcpMethod(method)
    vm = VoxelHashMap(1.0, 30, ctx); vm.AddPoints(world)
    if method >= 2: vm.CalVoxelCovAll()
    hosts, T0s = [], []
    for i in range(300):
        n = int(rng.choice([0, 1, 300, 5000, 20000, 60000, 131072]))
        sc, Tt = synth.make_scan(world, max(n, 1), seed=10 + i)
        hosts.append(np.ascontiguousarray(sc[:n])); T0s.append(synth.perturb(Tt, seed=100 + i, max_trans=0.3, max_rot_deg=1.5))
    reg = Registration(RegistrationConfig(icp_method=m), ctx)
    scans = [Scan(ctx, h) for h in hosts]
    ref = reg.RunRegisterStream(scans, vm, T0s, slots=64)
    key = lambda out: np.concatenate([o["T"].ravel() for o in out] + [np.array([o["iterations"] for o in out], float)])
    kref = key(ref)
    pin = PinnedBuffer(max(1, sum(h.size for h in hosts)))
    packed = [reg.pack_host_inputs(hosts, T0s), reg.pack_host_inputs(hosts, T0s, pinned=pin)]
    for rep in range(reps):
        slots = [64, 7, 128, 3, 300, 33][rep % 6]
        out = reg.RunRegisterStreamHost(packed[rep % 2], vm, slots=slots)
        k = key(out)
        assert np.array_equal(k, kref, equal_nan=True), (method, rep, slots, np.nanmax(np.abs(k - kref)))
    print(m.name, reps, "host-fed runs (6 slot counts, pageable / page-locked): bit-identical to the resident stream")
print("host-fed soak OK")
