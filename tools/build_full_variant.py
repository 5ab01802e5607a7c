"""Whole-library A/B build: every source of the product compiled with extra flags into libbsc_amd/lib/variants/libbsc_NAME.so
(load with BSC_LIB_OVERRIDE=<path>).  For switches that touch several translation units (tools/build_variant.sh rebuilds one).
    python tools/build_full_variant.py NAME -DFOO=1 ..."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libbsc_amd import build as B
name, extra = sys.argv[1], sys.argv[2:]
out = os.path.join(B.HERE, "lib", "variants"); objd = os.path.join(out, "obj_" + name)
os.makedirs(objd, exist_ok=True)
def comp(src):
    obj = os.path.join(objd, os.path.splitext(os.path.basename(src))[0] + ".o")
    cmd = B._command(src, obj)
    cmd = cmd[:1] + extra + ["-DBSC_EXPERIMENT_BUILD"] + cmd[1:]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode: raise RuntimeError(r.stderr[-3000:])
    return obj
with ThreadPoolExecutor(6) as ex: objs = list(ex.map(comp, B._sources()))
lib = os.path.join(out, f"libbsc_{name}.so")
r = subprocess.run([B.HIPCC, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread", "-Wl,-Bsymbolic"], capture_output=True, text=True)
if r.returncode: raise RuntimeError(r.stderr[-3000:])
print("built", lib)
