"""Print VGPR/AGPR/scratch/LDS/occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import glob, os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tinympc-julia_amd", "csrc")
srcs = (sorted(glob.glob(os.path.join(here, "inst_*.hip"))) + sorted(glob.glob(os.path.join(here, "minst_*.hip"))) +
        [os.path.join(here, "sinst_g4_3.hip"), os.path.join(here, "kernels.hip")])  # + the stream kernels of the (6, x) shapes
def run(src):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-honor-nans",
                          "--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True).stderr
    rows, cur = [], {}
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPRs Spill"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and "remark" in line and key + ":" in line.split("remark:")[-1] and cur is not None:
                cur.setdefault(key, int(m.group(1)))
    return rows
with ThreadPoolExecutor(4) as ex:
    allrows = [r for rows in ex.map(run, srcs) for r in rows]
print(f"{'kernel':78s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'spill':>6s} {'LDS':>7s} {'occ':>4s}")
for r in allrows:
    n = r["name"].replace("tmpc::", "").replace("(tmpc::AdmmParams)", "").replace("void ", "")
    print(f"{n:78s} {r.get('VGPRs',0):5d} {r.get('AGPRs',0):5d} {r.get('ScratchSize [bytes/lane]',0):8d} {r.get('VGPRs Spill',0):6d} {r.get('LDS Size [bytes/block]',0):7d} {r.get('Occupancy [waves/SIMD]',0):4d}")
