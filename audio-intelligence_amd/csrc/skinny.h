// Shared between the decode (weight-streaming) GEMM kernels: gemm_skinny.hip (general form: f32 / bf16, M <= 64) and
// gemm_stream.hip (the persistent bf16 form the 7B decode step runs: M <= 16, K % 512 == 0).
#pragma once
#include "common.h"

struct SkinnyP {
    const char* A;
    const char* W;
    const char* bias;
    const char* res;
    const char* norm_w;
    char* C;
    int M, N, K;
    long long lda, ldw, ldc, ldres;
    int out_f32;
    float norm_eps;
    int swiglu_out;
    int a_rows;      // rows of the activation image kept in LDS (8 or 16, >= M)
    int tile_rows;   // weight rows per 16-wide MFMA tile that carry work (<= 16): narrow outputs are cut into ceil(N / CUs)-row shares so every CU streams the same bytes
    int n_tiles;     // gemm_stream.hip: 16-wide tiles per unit (1, 2 or 4)
    // gemm_stream.hip, lm_head of the greedy decode step: per-unit (value, index) partials of the first-index argmax over the allowed
    // intervals, logits rounded to bf16 first (llm.hip: masked_argmax).  NULL = off
    const int32_t* am_iv; int am_n_iv; float* am_val; int* am_idx;
#ifdef AFHIP_STREAM_STAMPS
    unsigned long long* dbg;   // diagnostic build only: 8 stamps per workgroup
#endif
};

enum { SKINNY_A_PLAIN = 0, SKINNY_A_RMSNORM = 1, SKINNY_A_SWIGLU = 2 };

// gemm_stream.hip: true when the persistent form took the launch (bf16, M <= 16, K % 512 == 0, the activation image or the per-wave
// staging slots fit in LDS); false = the caller falls back to skinny_kernel
bool afhip_gemm_stream_bf16(const SkinnyP& p, int amode, hipStream_t s);
