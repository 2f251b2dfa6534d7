"""Instruction census of the headline kernel's ADMM iteration from the compiler's own assembly (hipcc -S of csrc/linst_4_1_20.hip):
admm_lean_kernel<4,1,20, LIVE=false, UBK=true, ONE=true, XB=false, REFS=zero> — the benchmark's instantiation (tolerances <= 0, bounds constant over
the knots, one wavefront per SIMD) — its innermost loop (the iterations that do not form residuals: 99 of 100), counted by class,
next to round 3's census of quad<4,1,20,g1>.  Output: profiles/<tag>_cartpole_isa_census.json"""
import collections, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
asm = "/tmp/lean20_census.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-honor-nans", "-fno-slp-vectorize", "--cuda-device-only",
                "-I" + os.path.join(ROOT, "tinympc-julia_amd/csrc"), "-S", os.path.join(ROOT, "tinympc-julia_amd/csrc/linst_4_1_20.hip"), "-o", asm], check=True,
               stderr=subprocess.DEVNULL)
lines = open(asm).read().splitlines()
want = "_ZN4tmpc16admm_lean_kernelILi4ELi1ELi20ELb0ELb1ELb1ELb0ELi0EfEEvNS_10AdmmParamsE:"
start = next(i for i, l in enumerate(lines) if l.startswith(want))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
lab = {}
for n, b in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", b)
    if m and ("Inner Loop Header" in b or (n + 1 < len(body) and "Inner Loop Header" in body[n + 1])): lab[m.group(1)] = n
loops = []
for n, b in enumerate(body):
    m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", b)
    if m and m.group(1) in lab and lab[m.group(1)] < n: loops.append((lab[m.group(1)], n))
def is_inst(l):
    s = l.strip()
    return bool(s) and not s.startswith((";", ".", "//")) and not s.endswith(":")
lo, hi = max(loops, key=lambda lp: sum(1 for l in body[lp[0]:lp[1] + 1] if is_inst(l) and re.match(r"v_(fma|fmac|mul|add)_f64", l.strip())))
classes = collections.OrderedDict([
    ("fp64 FMA / mul / add (the recurrences)", r"v_(fma|fmac|mul|add)_f64"),
    ("fp32 <-> fp64 conversions", r"v_cvt_f(32_f64|64_f32)"),
    ("AGPR moves (v_accvgpr_read / write)", r"v_accvgpr_"),
    ("fp32 arithmetic (add / sub)", r"v_(add|sub|subrev|mul|fma|fmac)_f32"),
    ("fp32 med3 (the box projection)", r"v_(min|max|med3)_f32"),
    ("moves / selects / integer VALU", r"v_(mov|cndmask|readlane|readfirstlane|writelane|add_u32|lshl|and|or|cmp)"),
    ("other VALU", r"v_"),
    ("scalar memory", r"s_load|s_buffer_load"), ("scalar ALU / control", r"s_(?!waitcnt|nop|load|buffer_load)"), ("waits / nops", r"s_waitcnt|s_nop"),
    ("LDS", r"ds_"), ("global / scratch memory", r"global_|scratch_|buffer_|flat_")])
counts = collections.OrderedDict((k, 0) for k in classes)
for l in body[lo:hi + 1]:
    if not is_inst(l): continue
    op = l.strip().split()[0]
    for k, pat in classes.items():
        if re.match(pat, op): counts[k] += 1; break
valu = sum(v for k, v in counts.items() if k.split()[0] in ("fp64", "fp32", "AGPR", "moves", "other"))
meta = {}
for l in lines:
    pass
out = {"kernel": "lean<4,1,20> = admm_lean_kernel<4,1,20, LIVE=false, UBK=true, ONE=true, XB=false, REFS=zero> (the benchmark's instantiation)",
       "instructions_per_iteration": sum(counts.values()), "by_class": counts, "valu_instructions": valu,
       "necessary_fp64_fma": counts["fp64 FMA / mul / add (the recurrences)"],
       "necessary_share_of_valu": counts["fp64 FMA / mul / add (the recurrences)"] / valu,
       "floor_ms_100_iterations_at_4_cycles_2.4GHz": 4 * valu * 100 / 2.4e9 * 1e3,
       "round3_quad_g1": {"valu_instructions": 1506, "fp64": 911, "conversions": 228, "agpr_moves": 268, "kernel_ms": 0.3537},
       "measured": "profiles/r04_lean_timeline.txt: 4.09 core cycles per loop instruction in-kernel (s_memtime), 2.08 GHz with every CU busy (2.35 GHz at 80 CUs)"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_cartpole_isa_census.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
