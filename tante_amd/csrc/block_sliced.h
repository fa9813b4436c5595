// Internal interface of the feature-sliced fused TransformerBlock kernel (block_sliced.hip), called by the C-ABI entry points
// in block_fused.hip (tante_block_fused_supported / tante_block_stream_bytes / tante_pack_block / tante_block_fused).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tante_hip.h"

int tante_fs_supported(int C, int n_head, int hidden, int L, int causal);
int64_t tante_fs_stream_bytes(int C, int hidden);
void tante_fs_pack(const float* ln1_w, const float* ln1_b, const float* in_w, const float* in_b, const float* out_w, const float* out_b,
                   const float* ln2_w, const float* ln2_b, const float* fc1_w, const float* fc1_b, const float* fc2_w,
                   const float* fc2_b, char* dst, hipStream_t s);
void tante_fs_pack_folded(const float* in_w, const float* in_b, const float* out_w, const float* out_b, const float* fc1_w,
                          const float* fc1_b, const float* fc2_w, const float* fc2_b, char* dst, hipStream_t s);
constexpr int TANTE_FSP_MAX = 12;      // entries per tante_fs_pack_folded_multi launch
void tante_fs_pack_folded_multi(const float* const (*params)[8], char* const* dst, int n, hipStream_t s);
int tante_fs_launch(float* x, const char* stream, const TanteSeq& sq, int causal, float eps, hipStream_t s, const TanteBlockTrain* tr = nullptr,
                    const float* tprop = nullptr);
