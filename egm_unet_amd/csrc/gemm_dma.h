// Internal interface of csrc/gemm_dma.hip (the 8-wave LDS-DMA bf16 GEMM behind egm_gemm, csrc/vit.hip).
#pragma once
#include "common.h"

struct GemmDmaArgs {
    const void* A; const void* B; void* C; const float* bias; const void* R;      // bf16 A [M][K], B [N][K], C [M][N], R [M][N]; fp32 bias [N]
    int lda, ldb, ldc, ldr, M, N, K, act;
    float alpha;
};
// 1 when the product takes the LDS-DMA kernel (shape, alignment, enough tiles to fill the chip, EGM_GEMM_DMA != 0)
int egm_gemm_dma_ok(const GemmDmaArgs& a);
int egm_gemm_dma_launch(const GemmDmaArgs& a, hipStream_t st);
