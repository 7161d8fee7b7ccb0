"""Developer soak (GPU): every method, 25 continuous-batching runs with four slot counts + one lockstep batch over the same 48
registrations (large initial errors, so many points take the wave-cooperative stage) must be bit-identical.
    python tools/soak_determinism.py [--group 0,0,0]     --group: the same through a device group (elm_ctx_create_multi) over those devices"""
import sys, numpy as np
sys.path.insert(0, '.')
from elimaloc_amd import synth
from elimaloc_amd.registration import Context, VoxelHashMap, Registration, RegistrationConfig, IcpMethod, Scan
ctx = Context.multi([int(x) for x in sys.argv[sys.argv.index("--group") + 1].split(",")]) if "--group" in sys.argv else Context(0)
print("context:", ctx.group_info())
world = synth.make_world(2_000_000, seed=1001)
for method in (0, 1, 2, 3):
    m = IcpMethod(method)
    vm = VoxelHashMap(1.0, 30, ctx); vm.AddPoints(world)
    if method >= 2: vm.CalVoxelCovAll()
    if method == 1: vm.CalPointCovAll(0.4)
    scans, T0s = [], []
    for i in range(48):
        sc, Tt = synth.make_scan(world, 40000, seed=10 + i)
        scans.append(Scan(ctx, sc)); T0s.append(synth.perturb(Tt, seed=100 + i, max_trans=0.3, max_rot_deg=1.5))
    reg = Registration(RegistrationConfig(icp_method=m), ctx)
    ref = None
    for rep in range(25):
        slots = [16, 7, 48, 3][rep % 4]
        out = reg.RunRegisterStream(scans, vm, T0s, slots=slots)
        key = np.concatenate([o["T"].ravel() for o in out] + [np.array([o["iterations"] for o in out], float)])
        if ref is None: ref = key
        assert np.array_equal(key, ref), (method, rep, slots, np.abs(key - ref).max())
    lock = reg.RunRegisterBatch(scans, vm, T0s)
    key = np.concatenate([o["T"].ravel() for o in lock] + [np.array([o["iterations"] for o in lock], float)])
    assert np.array_equal(key, ref), (method, "lockstep")
    print(m.name, "25 stream runs with 4 slot counts + lockstep batch: bit-identical; iterations", sorted(set(int(o["iterations"]) for o in out)))
print("soak OK")
