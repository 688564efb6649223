// Persistent weight-streaming GEMM of the 7B decode step as ONE launch (bf16, M <= 16 rows, K % 512 == 0): C[M,N] = A'[M,K] . W[N,K]^T.
// Replaces, per decoder layer of ParallelLLM._step (lm/parallel.py:570-597 over modeling_qwen2.py:195-235, 35-49): q|k|v with the
// RMSNorm in front of it and its bias, the o projection + residual, gate/up with RMSNorm + SwiGLU, the down projection + residual;
// and the lm_head (lm/parallel.py:592).  The phase itself (weight window, activation image / slots, K-slice combine, epilogues) is
// stream_phase.h; the decode step itself runs img_phase.h (activations handed over as fragment-order images, decode_phases.hip).
#include "stream_phase.h"
#include <stdlib.h>

namespace {

using stream::NW;
using stream::KS;

template <int AMODE, int NT, bool PAIR, bool SLOT, int RM, int DEPTH>
__global__ __launch_bounds__(512) void skinny_stream_kernel(SkinnyP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    stream::StreamPhase<AMODE, NT, PAIR, SLOT, RM, DEPTH> ph(p, smem);
    ph.begin();
    ph.run();
}

constexpr size_t STREAM_LDS_MAX = 156 * 1024;

template <int AMODE, int NT, bool PAIR, bool SLOT, int RM, int DEPTH>
void launch(const SkinnyP& p, int grid, size_t lds, hipStream_t s) {
    static unsigned long long attr_done = 0;
    if (afhip_first_use_on_device(&attr_done))
        (void)hipFuncSetAttribute((const void*)skinny_stream_kernel<AMODE, NT, PAIR, SLOT, RM, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)STREAM_LDS_MAX);
    hipLaunchKernelGGL((skinny_stream_kernel<AMODE, NT, PAIR, SLOT, RM, DEPTH>), dim3((unsigned)grid), dim3(512), lds, s, p);
}

template <int NT, bool PAIR, int RM>
bool launch_rm(const SkinnyP& p, int amode, int grid, hipStream_t s) {
    // K steps in flight per wave.  The narrow projections (one unit per workgroup, 7 steps per wave at K = 3584) put a wave's whole K
    // share in flight at once: a window of 4 pays a second HBM round trip for steps 4..6.  Four tiles per step: 2 (64 weight registers)
    constexpr int D = (NT <= 2 && !PAIR) ? 7 : (NT == 4 ? 2 : 4);
    const size_t fixed = stream::StreamPhase<SKINNY_A_PLAIN, NT, PAIR, false, RM, D>::fixed_lds();
    const size_t img = (size_t)RM * p.K * 2;
    if (fixed + img <= STREAM_LDS_MAX) {
        if (amode == SKINNY_A_RMSNORM) launch<SKINNY_A_RMSNORM, NT, PAIR, false, RM, D>(p, grid, fixed + img, s);
        else launch<SKINNY_A_PLAIN, NT, PAIR, false, RM, D>(p, grid, fixed + img, s);
        return true;
    }
    if (amode != SKINNY_A_PLAIN) return false;
    launch<SKINNY_A_PLAIN, NT, PAIR, true, RM, D>(p, grid, fixed + (size_t)NW * D * RM * 128, s);
    return true;
}

template <int NT, bool PAIR>
bool launch_nt(const SkinnyP& p, int amode, int grid, hipStream_t s) {
    return p.a_rows == 8 ? launch_rm<NT, PAIR, 8>(p, amode, grid, s) : launch_rm<NT, PAIR, 16>(p, amode, grid, s);
}

}  // namespace

bool afhip_gemm_stream_bf16(const SkinnyP& p_in, int amode, hipStream_t s) {
    SkinnyP p = p_in;
#ifdef AFHIP_STREAM_STAMPS
    { const char* dp = getenv("AFHIP_STREAM_DBGPTR"); p.dbg = dp ? (unsigned long long*)strtoull(dp, nullptr, 0) : nullptr; }
#endif
    if (p.M > 16 || p.K % (KS * NW) != 0 || amode == SKINNY_A_SWIGLU) return false;
    const int cus = afhip_cu_count();
    if (p.swiglu_out) {
        const int units = cdiv(p.N / 2, p.tile_rows);
        return launch_nt<2, true>(p, amode, units < cus ? units : cus, s);
    }
    const int units = cdiv(p.N, p.n_tiles * p.tile_rows);
    const int grid = units < cus ? units : cus;
    switch (p.n_tiles) {
        case 1: return launch_nt<1, false>(p, amode, grid, s);
        case 2: return launch_nt<2, false>(p, amode, grid, s);
        case 4: return launch_nt<4, false>(p, amode, grid, s);
        default: return false;
    }
}
