// api.hip -- version / error strings of the C ABI.
#include "vpf_common.h"
#include <stdlib.h>
#include <string.h>
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
        case 7: return (int)sizeof(VpfCaFront);
        default: return -1;
    }
}

// ------------------------------------------------------------------ VpfDebug: the one place that reads the environment
struct VpfDebugKey { const char* name; const char* env; int VpfDebug::*field; int dflt; };
static const VpfDebugKey kDebugKeys[] = {
    {"attn_resident", "VPF_ATTN_RESIDENT", &VpfDebug::attn_resident, 1},
    {"g2e_grid", "VPF_G2E_GRID", &VpfDebug::g2e_grid, 256},
    {"g2e_w4_grid", "VPF_G2E_W4_GRID", &VpfDebug::g2e_w4_grid, 0},
    {"wgrad_cfg", "VPF_WGRAD_CFG", &VpfDebug::wgrad_cfg, 0},
    {"wgrad_wgs", "VPF_WGRAD_WGS", &VpfDebug::wgrad_wgs, 0},
    {"gemm_cfg", "VPF_GEMM_CFG", &VpfDebug::gemm_cfg, -1},
    {"wgroup_cfg", "VPF_WGROUP_CFG", &VpfDebug::wgroup_cfg, 2},
    {"wgroup_wgs", "VPF_WGROUP_WGS", &VpfDebug::wgroup_wgs, 0},
    {"wgroup_uneven", "VPF_WGROUP_UNEVEN", &VpfDebug::wgroup_uneven, 2},
    {"wgroup_dbg", "VPF_WGROUP_DBG", &VpfDebug::wgroup_dbg, 0},
    {"fps_exclusive_cu", "VPF_FPS_EXCLUSIVE_CU", &VpfDebug::fps_exclusive_cu, 0},
    {"knn_select", "VPF_KNN_SELECT", &VpfDebug::knn_select, 1},
    {"sa_nj", "VPF_SA_NJ", &VpfDebug::sa_nj, 1},
    {"sa_bwd_rows", "VPF_SA_BWD_ROWS", &VpfDebug::sa_bwd_rows, 1},
    {"smallk_rpb", "VPF_SMALLK_RPB", &VpfDebug::smallk_rpb, 0},
    {"sa_wg2", "VPF_SA_WG2", &VpfDebug::sa_wg2, 0},
    {"attn_ksplit", "VPF_ATTN_KSPLIT", &VpfDebug::attn_ksplit, 4},
    {"attn_ca_merged", "VPF_ATTN_CA_MERGED", &VpfDebug::attn_ca_merged, 128},
    {"attn_rng32", "VPF_ATTN_RNG32", &VpfDebug::attn_rng32, 1},
    {"sa_bwd_fuse", "VPF_SA_BWD_FUSE", &VpfDebug::sa_bwd_fuse, 1},
    {"wgroup_xlist", "VPF_WGROUP_XLIST", &VpfDebug::wgroup_xlist, 1},
    {"sa_rb", "VPF_SA_RB", &VpfDebug::sa_rb, 0},
    {"sa_stagger", "VPF_SA_STAGGER", &VpfDebug::sa_stagger, 0},
    {"sa_store", "VPF_SA_STORE", &VpfDebug::sa_store, 0},
    {"sa_tpw", "VPF_SA_TPW", &VpfDebug::sa_tpw, 0},
    {"wgroup_dma", "VPF_WGROUP_DMA", &VpfDebug::wgroup_dma, 2048},
    {"wgroup_dma_tn", "VPF_WGROUP_DMA_TN", &VpfDebug::wgroup_dma_tn, 128},
    {"wgroup_dma_ramp", "VPF_WGROUP_DMA_RAMP", &VpfDebug::wgroup_dma_ramp, 50},
};
VpfDebug& vpf_debug()
{
    static VpfDebug d = [] {
        VpfDebug v;
        for (const VpfDebugKey& k : kDebugKeys) {
            const char* e = getenv(k.env);
            v.*(k.field) = e ? atoi(e) : k.dflt;
        }
        return v;
    }();
    return d;
}
extern "C" int vpf_operand_dtype(void) { return VPF_OPERAND_FP16 ? 1 : 0; }
extern "C" int vpf_debug_set(const char* key, int value)
{
    if (!key) return VPF_ERR_NULL;
    for (const VpfDebugKey& k : kDebugKeys)
        if (!strcmp(k.name, key)) { vpf_debug().*(k.field) = value; return VPF_OK; }
    return VPF_ERR_UNSUPPORTED;
}
extern "C" int vpf_debug_get(const char* key, int* value)
{
    if (!key || !value) return VPF_ERR_NULL;
    for (const VpfDebugKey& k : kDebugKeys)
        if (!strcmp(k.name, key)) { *value = vpf_debug().*(k.field); return VPF_OK; }
    return VPF_ERR_UNSUPPORTED;
}
