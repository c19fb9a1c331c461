#!/usr/bin/env python3
"""A/B library variants on one box: tools/mkvariant.py <name> [file.hip=+flag,-flag ...] [-DX=1 ...]
Compiles the named files with flags added (+) / removed (-) from the library's normal flags (bare -D... arguments go to every NAMED
file), links them with the normal objects of the rest into tools/_bin/libvipformer_<name>.so (git-ignored, travels with the gpurun
snapshot) and prints the packed-fp32 count per changed file.  Use: VPF_LIB=$PWD/tools/_bin/libvipformer_<name>.so python bench.py ...
The variant carries the normal build id (it is an experiment, never the shipped library)."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from vipformer_amd import build as B

name = sys.argv[1]
per, defs = {}, []
for a in sys.argv[2:]:
    if a.startswith("-D"):
        defs.append(a)
    else:
        f, _, fl = a.partition("=")
        per[f] = [x for x in fl.split(",") if x]
B.build(verbose=False)
out = os.path.join(root, "tools", "_bin"); os.makedirs(out, exist_ok=True)
objdir = os.path.join(out, "_obj_" + name); os.makedirs(objdir, exist_ok=True)
objs = []
for f in sorted(x for x in os.listdir(B.CSRC) if x.endswith(".hip")):
    if f not in per:
        objs.append(os.path.join(B.OBJ, f[:-4] + ".o")); continue
    flags = list(B.COMMON) + B.PER_FILE.get(f, [])
    for x in per[f]:
        if x.startswith("-"):
            flags = [c for c in flags if c != x[1:] and c != "-" + x[1:]]
        else:
            flags.append(x[1:] if x.startswith("+") else x)
    if f == "api.hip":
        flags.append('-DVPF_BUILD_ID="' + B.source_hash() + '"')
    obj = os.path.join(objdir, f[:-4] + ".o")
    r = subprocess.run(["hipcc"] + flags + defs + ["-c", os.path.join(B.CSRC, f), "-o", obj], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-4000:])
    n = 0
    for img in B.device_code_objects(obj):
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".co") as t:
            t.write(img); t.flush()
            d = subprocess.run([B.OBJDUMP, "-d", t.name], capture_output=True, text=True).stdout
        n += len(re.findall(r"\bv_pk_(?:add|mul|fma)_f32\b", d))
    print(f"{f}: {' '.join(per[f] + defs)} -> {n} packed-fp32 instructions")
    objs.append(obj)
lib = os.path.join(out, f"libvipformer_{name}.so")
r = subprocess.run(["hipcc", "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-4000:])
print(lib)
