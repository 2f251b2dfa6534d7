"""Register / scratch / LDS / occupancy table of EVERY kernel in the built library, from the code objects themselves (the
AMDGPU metadata notes of the gfx950 code objects embedded in lib/libtinympc_hip.so: .vgpr_count, .agpr_count,
.vgpr_spill_count, .private_segment_fixed_size, .group_segment_fixed_size) — seconds, no recompilation, and by construction
the table of the library that ships.  usage: kernel_resources_so.py [lib.so] > profiles/rNN_kernel_resources.txt"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tinympc-julia_amd", "lib", "libtinympc_hip.so")
rows = []
with tempfile.TemporaryDirectory() as td:
    # the embedded clang offload bundles: magic, u64 entry count, then per entry (u64 offset, u64 size, u64 triple length, triple)
    import struct
    blob = open(lib, "rb").read()
    MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
    pieces, pos = [], blob.find(MAGIC)
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size > 0: pieces.append((pos + off, size))
        pos = blob.find(MAGIC, pos + 1)
    for i, (off, size) in enumerate(pieces):
        co = os.path.join(td, f"co{i}.o")
        with open(co, "wb") as g:
            g.write(blob[off:off + size])
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        cur = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                cur = {"sym": m.group(1)}
                rows.append(cur)
            for key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size"):
                m = re.match(r"\s+\.%s:\s+(\d+)" % key, line)
                if m and cur is not None: cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(r["sym"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print(f"# {len(rows)} kernels in {os.path.relpath(lib, ROOT)} ({os.path.getsize(lib) / 1e6:.1f} MB); VGPR = architectural, AGPR = accumulation registers, "
      f"occ = wavefronts per SIMD by registers (512 / (VGPR + AGPR), granule 8), LDS and scratch in bytes per workgroup / lane")
print(f"{'kernel':118s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spillV':>7s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for r, n in sorted(zip(rows, names), key=lambda rn: rn[1]):
    n = n.replace("tmpc::", "").replace("(AdmmParams)", "").replace("void ", "")
    tot = (r.get("vgpr_count", 0) + 7) // 8 * 8 + (r.get("agpr_count", 0) + 7) // 8 * 8
    occ = max(1, min(8, 512 // max(tot, 1)))
    print(f"{n[:118]:118s} {r.get('vgpr_count', 0):5d} {r.get('agpr_count', 0):5d} {r.get('sgpr_count', 0):5d} {r.get('vgpr_spill_count', 0):7d} "
          f"{r.get('private_segment_fixed_size', 0):8d} {r.get('group_segment_fixed_size', 0):7d} {occ:4d}")
