// Sampling branch of the UALM decode loop on gfx950: classifier-free-guidance mix + re-mask, top-k, temperature softmax and an
// inverse-CDF draw, one workgroup per (sequence, stream) row.  Mirrors lm/parallel.py:489-492 (logits * cfg + cfg_logits * (1 - cfg),
// masked_fill_ with the modality mask) and :603-608 (topk -> softmax(values / temperature) -> multinomial -> gather).
// The multinomial draw is taken from a caller-supplied uniform per row (same distribution, no RNG stream to reproduce).
#include "common.h"

namespace {

constexpr int SK_MAX = 64;

struct SampleP {
    const float* logits;
    const float* cfg_logits;
    float cfg, one_minus_cfg;
    int rows, ld;
    const int32_t* allowed;
    int n_iv, k;
    float temperature;
    int round_bf16;
    int32_t* topk_idx;
    float* topk_val;
    float* topk_prob;
    const float* u;
    int64_t* token;
};

// the value the reference's tensors hold at (row, id): model-dtype logits, mixed with three separately rounded tensor ops
template <bool RB>
__device__ __forceinline__ float mixed_value(const SampleP& p, const float* row, const float* crow, int i) {
#pragma clang fp contract(off)
    float v = row[i];
    if (RB) v = (float)(bf16)v;
    if (crow) {
        float c = crow[i];
        if (RB) c = (float)(bf16)c;
        float a = v * p.cfg;
        float b = c * p.one_minus_cfg;
        if (RB) { a = (float)(bf16)a; b = (float)(bf16)b; }
        v = a + b;
        if (RB) v = (float)(bf16)v;
    }
    return v;
}

// (v, i) precedes (w, j) in the output order: larger value first, ties by smaller id
__device__ __forceinline__ bool before(float v, int i, float w, int j) { return v > w || (v == w && i < j); }

template <bool RB>
__global__ __launch_bounds__(256) void sample_topk_kernel(SampleP p) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float kv[SK_MAX];
    __shared__ int ki[SK_MAX];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* row = p.logits + (long long)r * p.ld;
    const float* crow = p.cfg_logits ? p.cfg_logits + (long long)r * p.ld : nullptr;
    const int32_t* iv = p.allowed + (long long)r * p.n_iv * 2;
    float pv = INFINITY;       // previous pick: everything strictly after it in the order is still a candidate
    int pi = -1;
    for (int j = 0; j < p.k; ++j) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int q = 0; q < p.n_iv; ++q) {
            const int lo = iv[2 * q], hi = iv[2 * q + 1];
            for (int i = lo + tid; i < hi; i += 256) {
                const float v = mixed_value<RB>(p, row, crow, i);
                if (before(pv, pi, v, i) && before(v, i, best, bi)) { best = v; bi = i; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (before(ov, oi, best, bi)) { best = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
        __syncthreads();
        for (int w = 0; w < 4; ++w)
            if (before(sv[w], si[w], best, bi)) { best = sv[w]; bi = si[w]; }
        // best / bi are now identical in every thread
        if (tid == 0) { kv[j] = best; ki[j] = bi; }
        pv = best;
        pi = bi;
        __syncthreads();
        if (bi == 0x7fffffff) {            // candidates exhausted (fewer than k finite allowed ids): pad like topk's -inf tail
            if (tid == 0)
                for (int t = j; t < p.k; ++t) { kv[t] = -INFINITY; ki[t] = iv[0]; }
            __syncthreads();
            break;
        }
    }
    if (tid == 0) {
        // softmax(values / temperature) over the k picks (lm/parallel.py:604); kv[0] is the maximum
        float e[SK_MAX];
        float sum = 0.f;
        const float x0 = kv[0] / p.temperature;
        for (int j = 0; j < p.k; ++j) {
            e[j] = kv[j] == -INFINITY ? 0.f : expf(kv[j] / p.temperature - x0);
            sum += e[j];
        }
        float cdf = 0.f;
        int pick = p.k - 1;
        bool done = false;
        const float uu = p.u ? p.u[r] : 0.f;
        for (int j = 0; j < p.k; ++j) {
            const float pr = sum > 0.f ? e[j] / sum : 0.f;
            if (p.topk_idx) p.topk_idx[(long long)r * p.k + j] = ki[j];
            if (p.topk_val) p.topk_val[(long long)r * p.k + j] = kv[j];
            if (p.topk_prob) p.topk_prob[(long long)r * p.k + j] = pr;
            cdf += pr;
            if (!done && uu < cdf) { pick = j; done = true; }
        }
        while (pick > 0 && e[pick] == 0.f) --pick;      // never land on a padded slot through rounding of the last cdf step
        if (p.token) p.token[r] = ki[pick];
    }
}

}  // namespace

extern "C" int afhip_sample_topk(const afhip_sample_args* a, void* stream) {
    AFHIP_CHECK(a != nullptr, "afhip_sample_topk: null args");
    AFHIP_CHECK(a->logits && a->allowed && a->rows > 0 && a->ld > 0 && a->n_iv > 0, "afhip_sample_topk: bad logits / allowed / shape");
    AFHIP_CHECK(a->k >= 1 && a->k <= SK_MAX, "afhip_sample_topk: k=%d must be in [1,%d]", a->k, SK_MAX);
    AFHIP_CHECK(a->temperature > 0.f, "afhip_sample_topk: temperature must be > 0 (greedy is afhip_masked_argmax)");
    AFHIP_CHECK(a->model_dtype == AFHIP_F32 || a->model_dtype == AFHIP_BF16, "afhip_sample_topk: bad model_dtype");
    AFHIP_CHECK(a->token == nullptr || a->u != nullptr, "afhip_sample_topk: token output needs the uniforms u");
    SampleP p;
    p.logits = a->logits; p.cfg_logits = a->cfg_logits; p.cfg = a->cfg; p.one_minus_cfg = a->one_minus_cfg != 0.f ? a->one_minus_cfg : (float)(1.0 - (double)a->cfg);
    p.rows = a->rows; p.ld = a->ld; p.allowed = a->allowed; p.n_iv = a->n_iv; p.k = a->k; p.temperature = a->temperature;
    p.round_bf16 = a->model_dtype == AFHIP_BF16;
    p.topk_idx = a->topk_idx; p.topk_val = a->topk_val; p.topk_prob = a->topk_prob; p.u = a->u; p.token = a->token;
    if (p.round_bf16) hipLaunchKernelGGL(sample_topk_kernel<true>, dim3(a->rows), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(sample_topk_kernel<false>, dim3(a->rows), dim3(256), 0, (hipStream_t)stream, p);
    AFHIP_LAUNCH_CHECK();
    return 0;
}
