// decode_chain.hip <-> llm.hip (C++ linkage, internal to the library)
#pragma once
#include "common.h"

// bf16 weights, B <= 16, widths the persistent phases take (stream_phase.h); AFHIP_DECODE_CHAIN=0 turns the chains off
bool afhip_decode_chain_supported(const afhip_llm_weights* w, int B);
// barrier words (first 2048 bytes: zeroed by the caller once per decode step) + argmax partials
size_t afhip_decode_chain_scratch_bytes(int B);

struct afhip_chain_step {            // one launch of a decode step
    const afhip_llm_weights* w;
    int B, layer;                    // layer l: o_l, gate/up_l, down_l, then q|k|v_{l+1} or (last layer) lm_head + pick; layer = -1: embed + q|k|v_0
    char* x; char* qkv; char* att; char* act;      // [B, hidden], [B, (n_q + 2 n_kv) hd], [B, n_q hd], [B, inter]
    void* scratch;
    const afhip_decode_state* st;
    int step;
    int bar0;                        // barrier rounds of the earlier launches of this step
};
// rounds_out: barrier rounds this launch adds
int afhip_decode_chain_launch(const afhip_chain_step& c, hipStream_t s, int* rounds_out);
