#!/usr/bin/env python3
"""Static VALU opcode mix of one kernel of the product library (device assembly from hipcc -S), classified by the issue cost the
VALU probe measured (profiles/r04_valu_probe.txt): which share of the kernel's vector instructions occupies its SIMD for ~2.4 cycles
and which for ~4.3.  Used to turn SQ_INSTS_VALU per SIMD-cycle into a pipe utilisation (bench.py roofline).
    python tools/valu_mix.py [mangled-name-regex] [--asm file.s]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else r"k_accumulate_gridILi0ELi0ELi0ELi0ELi0E"
asm = sys.argv[sys.argv.index("--asm") + 1] if "--asm" in sys.argv else None
if not asm:
    # the translation unit that holds the kernel (round 6: the library is seven of them)
    unit = "elm_k_vnbr.hip" if "vnbr" in pat else "elm_k_cell.hip" if "cell" in pat else "elm_k_solve.hip" if "solve" in pat else "elm_k_grid.hip"
    asm = os.path.join(tempfile.gettempdir(), unit.replace(".hip", "_gfx950.s"))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "elimaloc_amd", "csrc", unit), "-o", asm],
                          stderr=subprocess.DEVNULL)
txt = open(asm).read()
m = re.search(r"^(_Z\w*" + pat + r"\w*):[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M)
if not m:
    raise SystemExit("kernel not found")
# measured at 8 waves / SIMD (shader cycles per wave64 instruction): FAST ~2.4: v_fma_f32 v_add_f32 v_add_u32 v_and_b32 v_mov_b32;
# SLOW ~4.3: packed f32, every f64 op, v_and_or_b32 v_med3_u32 v_min_u32 v_max_f32 v_lshl_add_u32 v_lshlrev_b32 v_bfe_u32 v_mul_lo_u32
# v_cvt_* v_cmp_* v_cndmask_b32; TRANS ~8.2: v_rcp_f32 v_sqrt_f32.  Opcodes the probe did not time are classed with their family and
# listed under "assumed".
FAST = {"v_fma_f32", "v_add_f32", "v_add_u32", "v_and_b32", "v_mov_b32"}
FAST_ASSUMED = {"v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_sub_u32", "v_subrev_u32", "v_or_b32", "v_xor_b32", "v_not_b32",
                "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subrev_co_u32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
ops = collections.Counter()
for line in m.group(2).splitlines():
    line = line.strip()
    if not line or line[0] in ".;/" or line.endswith(":"):
        continue
    op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", line.split()[0])
    if op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        ops[op] += 1
tot = sum(ops.values())
cls = collections.Counter()
assumed = collections.Counter()
for op, c in ops.items():
    if op in FAST:
        cls["fast"] += c
    elif op in FAST_ASSUMED:
        cls["fast"] += c; assumed[op] += c
    elif op.startswith(TRANS):
        cls["trans"] += c
    else:
        cls["slow"] += c
print(f"{m.group(1)}")
print(f"static VALU instructions {tot}: fast (~2.4 cyc) {cls['fast']} = {cls['fast'] / tot:.3f}, slow (~4.3 cyc) {cls['slow']} = {cls['slow'] / tot:.3f}, "
      f"transcendental (~8.2 cyc) {cls['trans']} = {cls['trans'] / tot:.3f}")
print(f"mean issue cost {(2.4 * cls['fast'] + 4.3 * cls['slow'] + 8.2 * cls['trans']) / tot:.3f} cycles per wave64 instruction "
      f"(if every fast-assumed opcode were slow: {(2.4 * (cls['fast'] - sum(assumed.values())) + 4.3 * (cls['slow'] + sum(assumed.values())) + 8.2 * cls['trans']) / tot:.3f})")
print("assumed fast (not timed by the probe):", dict(assumed))
for op, c in ops.most_common(40):
    k = "fast" if op in FAST or op in FAST_ASSUMED else ("trans" if op.startswith(TRANS) else "slow")
    print(f"  {op:26s}{c:5d}  {k}")
