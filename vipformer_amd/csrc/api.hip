// api.hip -- version / error strings of the C ABI.
#include "vpf_common.h"
#include "vipformer_hip.h"

extern "C" int vpf_version(void)
{
    return 200;   // 0.2.0
}

// sha256 over every source of the library (csrc/*.hip, csrc/*.h, include/*.h), injected by vipformer_amd/build.py: build() compares it
// with the tree and rebuilds on a mismatch, so a prebuilt .so can never stand in for sources it was not compiled from.
#ifndef VPF_BUILD_ID
#define VPF_BUILD_ID "unknown"
#endif
extern "C" const char* vpf_build_id(void)
{
    return "VPF_BUILD_ID=" VPF_BUILD_ID;
}

extern "C" const char* vpf_strerror(int code)
{
    switch (code) {
        case VPF_OK: return "ok";
        case VPF_ERR_BADSHAPE: return "bad shape / size out of the supported range";
        case VPF_ERR_BADALIGN: return "pointer or leading dimension not aligned as required";
        case VPF_ERR_UNSUPPORTED: return "unsupported configuration";
        case VPF_ERR_HIP: return "HIP launch error";
        case VPF_ERR_NULL: return "null pointer for a required argument";
        default: return "unknown error";
    }
}

// sizeof of the argument structs of the ABI (a binding checks its own layout against the library's)
extern "C" int vpf_abi_sizeof(int which)
{
    switch (which) {
        case 0: return (int)sizeof(VpfPackJob);
        case 1: return (int)sizeof(VpfSaLayerFwd);
        case 2: return (int)sizeof(VpfWgradJob);
        case 3: return (int)sizeof(VpfSaLayerBwd);
        case 4: return (int)sizeof(VpfPgradJob);
        case 5: return (int)sizeof(VpfAdapterKv);
        case 6: return (int)sizeof(VpfAdapterKvBwd);
        default: return -1;
    }
}
