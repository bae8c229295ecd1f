// Internal launch interface of attention_kernel.hip (used by encoder.cpp; the public entry points are in proqa_hip.h).
#pragma once
#include <stdint.h>

namespace proqa {

// qkv_bias: [3*hidden] fp16 bias of the fused Q|K|V projection added inside the kernel, or nullptr when qkv
// already includes it.  Exactly one of seq_lens_dev (padded layout) / cu_seqlens_dev (packed layout) is non-null.
int launch_attention(const void* qkv, const void* qkv_bias, const int32_t* seq_lens_dev, const int32_t* cu_seqlens_dev,
                     int batch, int seq_len, int n_heads, void* ctx_out, void* stream);
int launch_attention_cls(const void* qkv, const void* qkv_bias, const int32_t* seq_lens_dev,
                         const int32_t* cu_seqlens_dev, int batch, int seq_len, int n_heads, void* ctx_cls_out,
                         void* stream);

}  // namespace proqa
