// crh_encoder.hip -- UniXcoder encoder entry points (placeholder bodies until the kernels land).
#include "crh_common.h"

extern "C" {
int crh_gemm_bf16_bias(const void *, const void *, const float *, void *, int, int, int, int, void *)
{
    return crh::fail(CRH_E_INTERNAL, "crh_gemm_bf16_bias: not implemented in this build");
}
int crh_gemm_bf16_bias_res_ln(const void *, const void *, const float *, const void *, const float *, const float *, float, void *, int,
                              int, int, void *)
{
    return crh::fail(CRH_E_INTERNAL, "crh_gemm_bf16_bias_res_ln: not implemented in this build");
}
int crh_attn_fwd_varlen(const void *, const int32_t *, void *, int, int, int, void *)
{
    return crh::fail(CRH_E_INTERNAL, "crh_attn_fwd_varlen: not implemented in this build");
}
int crh_embed_ln(const int32_t *, const void *, const void *, const void *, const float *, const float *, float, int, void *, int32_t *,
                 int, int, int, void *)
{
    return crh::fail(CRH_E_INTERNAL, "crh_embed_ln: not implemented in this build");
}
int crh_masked_mean_pool(const void *, const int32_t *, float *, int, int, int, void *)
{
    return crh::fail(CRH_E_INTERNAL, "crh_masked_mean_pool: not implemented in this build");
}
}
