"""Per-kernel resource + instruction summary of a hipcc -S assembly file: registers, scratch (spills), LDS and an
instruction-class histogram of the whole kernel body (static counts).  usage: asm_summary.py file.s [name-filter]"""
import collections, re, sys
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
lines = open(path).read().splitlines()
meta = {}
cur = None
for l in lines:
    m = re.match(r"\s+\.name:\s+(\S+)", l)
    if m: cur = m.group(1); meta.setdefault(cur, {})
    for key in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
        m = re.match(r"\s+\.%s:\s+(\d+)" % key, l)
        if m and cur: meta[cur][key] = int(m.group(1))
classes = [("fp64", r"v_(fma|fmac|mul|add|max|min)_f64"), ("cvt", r"v_cvt_f(32_f64|64_f32)"), ("accvgpr", r"v_accvgpr_"),
           ("fp32", r"v_(add|sub|subrev|mul|fma|fmac|mac|min|max|med3|max3|min3)_f32"), ("vmov", r"v_mov"), ("valu_other", r"v_"),
           ("scratch", r"scratch_"), ("lds", r"ds_"), ("global", r"global_|flat_|buffer_"), ("smem", r"s_load|s_buffer"),
           ("wait", r"s_waitcnt|s_nop"), ("salu", r"s_")]
start = None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\S+):", l)
    if m: start, name = i, m.group(1)
    if l.startswith(".Lfunc_end") and start is not None:
        if flt in name:
            c = collections.Counter()
            for b in lines[start:i]:
                s = b.strip()
                if not s or s.startswith((";", ".", "//")) or s.endswith(":"): continue
                op = s.split()[0]
                for k, pat in classes:
                    if re.match(pat, op): c[k] += 1; break
                else: c["other"] += 1
            print(name)
            print("   ", meta.get(name, {}))
            print("   ", dict(c), "total", sum(c.values()))
            # innermost loops (backward branches), largest first
            body = lines[start:i]
            # loop headers as the compiler marks them (the comment sits on the label's line or the one after)
            lab = {}
            for n, b in enumerate(body):
                m = re.match(r"^(\.LBB\d+_\d+):", b)
                if m and ("Inner Loop Header" in b or (n + 1 < len(body) and "Inner Loop Header" in body[n + 1])): lab[m.group(1)] = n
            inner = []
            for n, b in enumerate(body):
                m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", b)
                if m and m.group(1) in lab and lab[m.group(1)] < n: inner.append((lab[m.group(1)], n))
            for lo_, hi_ in sorted(inner, key=lambda lp: lp[0] - lp[1])[:6]:
                c2 = collections.Counter()
                for b in body[lo_:hi_ + 1]:
                    s2 = b.strip()
                    if not s2 or s2.startswith((";", ".", "//")) or s2.endswith(":"): continue
                    op = s2.split()[0]
                    for k, pat in classes:
                        if re.match(pat, op): c2[k] += 1; break
                    else: c2["other"] += 1
                valu = sum(v for k, v in c2.items() if k in ("fp64", "cvt", "accvgpr", "fp32", "vmov", "valu_other"))
                if c2.get("fp64", 0) > 100 or len(sys.argv) > 3:
                    print("    loop lines %d-%d:" % (lo_, hi_), dict(c2), "total", sum(c2.values()), "VALU", valu)
        start = None
