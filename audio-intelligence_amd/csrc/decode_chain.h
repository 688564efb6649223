// decode_chain.hip <-> llm.hip (C++ linkage, internal to the library)
#pragma once
#include "common.h"

// phases of a launch, executed in this order
enum { AFHIP_PH_EMBED = 1, AFHIP_PH_O = 2, AFHIP_PH_GU = 4, AFHIP_PH_DOWN = 8, AFHIP_PH_QKV = 16, AFHIP_PH_HEAD = 32, AFHIP_PH_PICK = 64 };

// AFHIP_DECODE_CHAIN: 0 = off (round-3 launches), 1 (default) = the imaged phases, one per launch, 2 = chained behind grid barriers
int afhip_decode_chain_mode();
// bf16 weights, B <= 16, widths the persistent phases take (img_phase.h)
bool afhip_decode_chain_supported(const afhip_llm_weights* w, int B);
// barrier words (first 2048 bytes: zeroed by the caller once per decode step), argmax partials, activation images
size_t afhip_decode_chain_scratch_bytes(const afhip_llm_weights* w, int B);
// the [8 or 16, n_q hd] fragment-order image the decode attention's merge writes and the o phase reads
void* afhip_decode_chain_att_image(const afhip_llm_weights* w, int B, void* scratch);

struct afhip_chain_step {            // one launch of a decode step
    const afhip_llm_weights* w;
    int B;
    int phases;                      // AFHIP_PH_* bits
    int layer;                       // of the o / gate-up / down phases
    int qkv_layer;                   // of the q|k|v phase
    char* x; char* qkv;              // plain rows: residual stream [B, hidden], q|k|v [B, (n_q + 2 n_kv) hd]
    void* scratch;
    const afhip_decode_state* st;
    int step;
    int bar0;                        // barrier rounds of the earlier launches of this step
};
// rounds_out: barrier rounds this launch adds
int afhip_decode_chain_launch(const afhip_chain_step& c, hipStream_t s, int* rounds_out);
