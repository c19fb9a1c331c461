"""In-tree build of libvipformer_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m vipformer_amd.build          # incremental
    python -m vipformer_amd.build --force

One object per .hip file (rebuilt when the source or a header is newer), linked into
vipformer_amd/libvipformer_hip.so.  The .so is git-ignored but travels to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libvipformer_hip.so")
ARCH = "gfx950"
# -fno-slp-vectorize: no packed-fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 ...) anywhere in the library.  With them fps_kernel
# mis-sampled whenever gemm_kernel workgroups shared its CU (the mechanism is not understood, so the
# instruction class goes everywhere, not only from the bit-exact kernels) -- and the step is 0.03 ms FASTER without them (A/B on one
# box, 4 alternating pairs: 4.296 vs 4.329 ms).
# -fno-vectorize (round 3): the LOOP vectoriser formed the same instructions in 14 small kernels (BatchNorm pieces, LayerNorm taps ...);
# check_no_packed_f32 below found them in the round-2 library that was believed to be free of them.
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-fno-slp-vectorize", "-fno-vectorize",
          "-I" + os.path.join(os.path.dirname(HERE), "include")]
# preproc.hip: bit-exact kernels -- no fused multiply-adds the source does not spell out
PER_FILE = {"preproc.hip": ["-ffp-contract=off"]}
# `python -m vipformer_amd.build --bf16` (or VPF_OPERAND=bf16 in the environment): the A/B build with bf16 MFMA operands (rounds 1-3)
# instead of fp16 -- csrc/vpf_common.h VPF_OPERAND_FP16; vipformer_amd._lib refuses a library that does not match VPF_OPERAND.
# VPF_EXTRA_CFLAGS="-DVPF_EXP_SETPRIO=1": experiment builds (part of the build id, so the library is rebuilt when it changes)
if os.environ.get("VPF_EXTRA_CFLAGS"):
    COMMON = COMMON[:8] + os.environ["VPF_EXTRA_CFLAGS"].split() + COMMON[8:]
if os.environ.get("VPF_OPERAND", "f16") == "bf16" or "--bf16" in sys.argv:
    COMMON = COMMON[:8] + ["-DVPF_OPERAND_FP16=0"] + COMMON[8:]


def source_hash() -> str:
    """sha256 over every source the library is compiled from (csrc/*.hip, csrc/*.h, include/*.h) and the compile flags."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "vipformer_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    h.update(" ".join([c for c in COMMON if not c.startswith("-I")] + sorted(sum(PER_FILE.values(), []))).encode())
    return h.hexdigest()


def built_hash() -> str:
    """The id embedded in the existing .so ('' if there is none): scanned from the file, nothing is loaded."""
    if not os.path.exists(LIB):
        return ""
    data = open(LIB, "rb").read()
    i = data.find(b"VPF_BUILD_ID=")
    return data[i + 13:i + 13 + 64].decode("ascii", "replace") if i >= 0 else ""


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "vipformer_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def device_code_objects(path: str):
    """The gfx950 code objects embedded in a host object / shared library: every AMDGPU ELF image found in the file."""
    import struct
    data = open(path, "rb").read()
    at = 0
    while True:
        i = data.find(b"\x7fELF\x02\x01\x01", at)
        if i < 0:
            return
        at = i + 4
        if len(data) < i + 64 or struct.unpack_from("<H", data, i + 18)[0] != 224:        # e_machine: EM_AMDGPU
            continue
        e_shoff, = struct.unpack_from("<Q", data, i + 40)
        e_shentsize, e_shnum = struct.unpack_from("<HH", data, i + 58)
        yield data[i:i + e_shoff + e_shentsize * e_shnum]


# s_load_dword* with BOTH a base pair and an offset register: the compiler forms it when it splits a run-time index into a by-value
# kernel-argument array of structs over the two (round 5: an 88-byte stride became base + 2 i, offset 86 i).  gfx950 drops the low two
# bits of each part, not of the sum -- odd indices read the neighbouring fields, silently.  No kernel of the library has one; the link
# step keeps it that way.  The ban is deliberately BROAD: any scalar load whose offset operand is a register -- sN, m0,
# vcc_lo / vcc_hi, ttmpN -- with or without an immediate is rejected, including the harmless dword-aligned `s_load_dword sX, s[a:b], sN`
# form; a kernel that trips it on a plain run-time index should take the index through a VGPR / global load instead (the error text says so).
import re as _re
SPLIT_SLOAD = _re.compile(r"^.*\bs_(?:buffer_)?load_dword(?:x\d+)?\s+s\[?[0-9:]+\]?,\s*s\[[0-9:]+\],\s*(?:s\d+|m0|vcc_lo|vcc_hi|ttmp\d+)\b.*$", _re.M)


def check_no_packed_f32(lib_path: str) -> int:
    """Fail the build if the library's device code contains a packed-fp32 VALU instruction (v_pk_add_f32 / v_pk_mul_f32 /
    v_pk_fma_f32).  With them fps_kernel mis-sampled whenever gemm_kernel workgroups shared its CU: the
    mechanism is not understood, so the instruction class is banned from the whole library -- -fno-slp-vectorize keeps the compiler
    from forming them, this check keeps a hand-written one (inline asm, a builtin, a vector type) from re-opening the fault
    silently.  The same pass rejects split scalar loads (SPLIT_SLOAD above).  Returns the number of code objects scanned."""
    import re
    import tempfile
    n, bad, bad_sload = 0, [], []
    for img in device_code_objects(lib_path):
        n += 1
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img); f.flush()
            d = subprocess.run([OBJDUMP, "-d", f.name], capture_output=True, text=True).stdout
        hits = re.findall(r"\bv_pk_(?:add|mul|fma)_f32\b", d)
        if hits:
            kernels = sorted(set(re.findall(r"^[0-9a-f]+ <([^>]+)>:", d, flags=re.M)))[:4]
            bad.append(f"code object {n}: {len(hits)} ({', '.join(kernels)} ...)")
        split = SPLIT_SLOAD.findall(d)
        if split:
            bad_sload.append(f"code object {n}: {len(split)} (first: {split[0].strip()})")
    if n == 0:
        raise RuntimeError(f"no gfx950 code object found in {lib_path}")
    if bad:
        raise RuntimeError("packed-fp32 instructions in the device code (the co-residency fault's trigger, DESIGN.md section 6): " + "; ".join(bad))
    if bad_sload:
        raise RuntimeError("scalar loads with a base AND an offset REGISTER in the device code (every such form is banned, also the dword-aligned one; "
                           "NOTES.md round 5: base + register + immediate loses the low two address bits of each part -- take the run-time index "
                           "through a VGPR / global load, or index the kernel-argument table another way): " + "; ".join(bad_sload))
    return n


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdr = _newest_header()
    want = source_hash()
    have = built_hash()
    if have != want and os.path.exists(LIB) and verbose:
        print(f"libvipformer_hip.so was built from other sources (id {have[:12] or 'none'} != {want[:12]}): recompiling every object")
    stale = have != want and os.path.exists(LIB)        # timestamps cannot be trusted for a library that travelled: rebuild all
    jobs = []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f[:-4] + ".o")
        extra = ['-DVPF_BUILD_ID="' + want + '"'] if f == "api.hip" else []
        if (force or stale or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr)
                or (f == "api.hip" and have != want)):
            jobs.append((["hipcc"] + COMMON + PER_FILE.get(f, []) + extra + ["-c", src, "-o", obj], f))

    def run(job):
        cmd, name = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {name}:\n{r.stderr[-6000:]}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr[-2000:])
        return name

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for name in ex.map(run, jobs):
                if verbose:
                    print("compiled", name)
    objs = [os.path.join(OBJ, f[:-4] + ".o") for f in srcs]
    if jobs or not os.path.exists(LIB) or force:
        r = subprocess.run(["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        n = check_no_packed_f32(LIB)
        if verbose:
            print(f"linked {LIB} ({n} code objects, no packed-fp32 instructions, no split scalar loads)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
