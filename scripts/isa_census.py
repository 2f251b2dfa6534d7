"""Instruction census of the headline kernel's ADMM iteration from the compiler's own assembly (hipcc -S of csrc/inst_4_1_20_g1.hip):
finds quad<4,1,20,g1>'s benched instantiation (zero references, fp64 recurrences, no finite state bound, one-shot loop), its
backward branches, takes the innermost loop with the most instructions (the iteration that does not report residuals — 99 of
100) and counts by class.  Output: profiles/<tag>_cartpole_isa_census.json + a text table."""
import json, os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
asm = sys.argv[2] if len(sys.argv) > 2 else "/tmp/q20.s"
if not os.path.isfile(asm):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "tinympc-julia_amd/csrc"),
                    "-fno-honor-nans", "--cuda-device-only", "-S", os.path.join(ROOT, "tinympc-julia_amd/csrc/inst_4_1_20_g1.hip"), "-o", asm], check=True)
lines = open(asm).read().splitlines()
want = "_ZN4tmpc16admm_quad_kernelINS_9QuadShapeILi4ELi1ELi20ELi1ELi520ELi520ELi3EEELi0EdLb0ELb1ELb1ELb0EEEvNS_10AdmmParamsE:"
start = next(i for i, l in enumerate(lines) if l.startswith(want))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
label_at = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in label_at and label_at[m.group(1)] < i:
        loops.append((label_at[m.group(1)], i))
def is_inst(l):
    s = l.strip()
    return bool(s) and not s.startswith((";", ".", "//")) and not s.endswith(":")
def count(lo, hi):
    return sum(1 for l in body[lo:hi + 1] if is_inst(l))
# innermost loops only (no other loop strictly inside)
inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
lo, hi = max(inner, key=lambda lp: count(*lp))
classes = collections.OrderedDict([
    ("fp64 FMA / mul / add (the recurrences)", r"v_(fma|fmac|mul|add)_f64"),
    ("fp32 <-> fp64 conversions", r"v_cvt_f(32_f64|64_f32)"),
    ("AGPR moves (v_accvgpr_read / write)", r"v_accvgpr_"),
    ("packed fp32 (v_pk_*)", r"v_pk_"),
    ("fp32 arithmetic (add / sub / mul / fma / mac)", r"v_(add|sub|subrev|mul|fma|fmac|mac|fmamk|fmaak)_f32"),
    ("fp32 min / max / med3 (box projections, residual maxima)", r"v_(min|max|med3|max3|min3)_f32"),
    ("moves / selects / integer VALU", r"v_(mov|cndmask|readlane|readfirstlane|writelane|add_u32|lshl|and|or|cmp)"),
    ("other VALU", r"v_"),
    ("scalar memory (coefficient fetch s_load)", r"s_load|s_buffer_load"),
    ("scalar ALU / control", r"s_(?!waitcnt|nop|load|buffer_load)"),
    ("waits / nops", r"s_waitcnt|s_nop"),
    ("LDS", r"ds_"),
    ("global / scratch memory", r"global_|scratch_|buffer_|flat_"),
])
counts = collections.OrderedDict((k, 0) for k in classes)
for l in body[lo:hi + 1]:
    if not is_inst(l):
        continue
    op = l.strip().split()[0]
    for k, pat in classes.items():
        if re.match(pat, op):
            counts[k] += 1
            break
    else:
        counts.setdefault("unclassified", 0)
        counts["unclassified"] += 1
total = sum(counts.values())
valu = sum(v for k, v in counts.items() if k.split()[0] in ("fp64", "fp32", "AGPR", "packed", "moves", "other"))
fma = counts["fp64 FMA / mul / add (the recurrences)"]
out = {"kernel": "quad<4,1,20,g1> (REFS zero, fp64 recurrences, XB off, one-shot loop)", "loop_lines": [lo, hi], "instructions_per_iteration": total,
       "by_class": counts, "valu_instructions": valu, "necessary_fp64_fma": fma,
       "floor_cycles_at_4_per_vector_instruction": 4 * valu,
       "floor_ms_100_iterations_at_2.4GHz": 4 * valu * 100 / 2.4e9 * 1e3,
       "note": "one wavefront per SIMD (the kernel's 470 values per lane leave no room for a second): the wavefront issues one vector "
               "instruction per 4 cycles at best, so 4 x valu_instructions is the floor of an iteration; the measured iteration "
               "(kernel time / 100 / cycle time) over this floor is the issue utilisation the SQ counters report"}
# what the counters say was EXECUTED per iteration (the loop carries both copies of the sweep — with and without the residual
# arithmetic — and branches over one of them): the latest committed SQ pass of the same kernel
import glob
for pth in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_cartpole_sq_counters.json")), reverse=True):
    e = json.load(open(pth))
    if e.get("kernel") == "quad<4,1,20,g1>" and e.get("SQ_INSTS_VALU") and e.get("SQ_WAVES"):
        ex = e["SQ_INSTS_VALU"] / e["SQ_WAVES"] / 100.0
        out["executed_valu_per_iteration_measured"] = ex
        out["executed_source"] = "profiles/" + os.path.basename(pth) + " (SQ_INSTS_VALU / SQ_WAVES / 100 iterations)"
        out["floor_ms_of_the_executed_count_at_4_cycles_2.4GHz"] = 4 * ex * 100 / 2.4e9 * 1e3
        break
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_cartpole_isa_census.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
