"""From-scratch build of the library in this container, for profiles/<tag>_build.json: `make clean`, then the same
`tinympc_julia_amd.build()` every loader runs (tests/conftest.py, __graft_entry__.build / smoke call ensure_built(), which
rebuilds whenever the hash of csrc/ + include/ differs from the stamp next to the .so)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
import tinympc_julia_amd as t
csrc = os.path.join(ROOT, "tinympc-julia_amd", "csrc")
subprocess.run(["make", "-s", "-C", csrc, "clean"], check=True)
if os.path.isfile(t.STAMP_PATH):
    os.remove(t.STAMP_PATH)
t0 = time.time()
t.build(jobs=8)
wall = time.time() - t0
stamp = json.load(open(t.STAMP_PATH))
lib = os.path.join(ROOT, "tinympc-julia_amd", "lib", "libtinympc_hip.so")
units = sorted(f for f in os.listdir(csrc) if f.endswith((".hip", ".cpp")))
out = {"what": "make clean + tinympc_julia_amd.build(jobs=8) in the build container (8 CPUs), gfx950 cross-compile",
       "wall_seconds": round(wall, 1), "stamp": stamp, "translation_units": len(units),
       "library_bytes": os.path.getsize(lib),
       "consumer_side": "tests/conftest.py::hip_lib, __graft_entry__.build() and smoke() call ensure_built(): the stamp's "
                        "source hash is compared with csrc/ + include/ and the library is rebuilt on a mismatch "
                        "(make is incremental: seconds when only one unit changed)"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_build.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
