// sa_rows.h -- launchers of the round-3 encoder row-block kernels (sa_rows.hip), called from the C-ABI entry points in sa_layer.hip.
#pragma once
#include "vpf_common.h"

// D = 256 (hidden 512, two workgroups per CU) or D = 384 (hidden 1536): VPF_OK, or VPF_ERR_UNSUPPORTED for any other width.
int sa_rows_fwd_launch(const VpfSaLayerFwd& a, hipStream_t st);
int sa_rows_bwd_mlp_launch(const VpfSaLayerBwd& a, hipStream_t st);
int sa_rows_bwd_qkv_launch(const VpfSaLayerBwd& a, hipStream_t st);
bool sa_rows_supported(int D, int hidden);
int sa_rows_bwd_pgrad_tokens(int D);
int sa_rows_ca_front_launch(const VpfCaFront& a, hipStream_t st);
int sa_rows_ca_front_bwd_launch(const VpfSaLayerBwd& a, hipStream_t st);          // D = 384 (32-token blocks); D = 256 has ca_front_bwd_rows_kernel
int sa_rows_adapter_kv_bwd_launch(const VpfAdapterKvBwd& a, hipStream_t st);      // D = 256 (two workgroups per CU) or 384
int sa_rows_adapter_kv_fwd_launch(const VpfAdapterKv& a, hipStream_t st);         // D = 384 (256 exists too; the round-2 kernel is the default there)
int sa_rows_adapter_kv_tokens(int D);                                             // tokens per workgroup = per kv-LayerNorm partial row of the backward kernel
