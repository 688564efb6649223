// decode_phases.hip <-> llm.hip (C++ linkage, internal to the library)
#pragma once
#include "common.h"

// the launches of a decode step
enum { AFHIP_PH_EMBED = 1, AFHIP_PH_O = 2, AFHIP_PH_GU = 4, AFHIP_PH_DOWN = 8, AFHIP_PH_QKV = 16, AFHIP_PH_HEAD = 32, AFHIP_PH_PICK = 64 };

// bf16 model, B <= 16, widths the persistent phases take (img_phase.h); option DECODE_IMAGED = 0 turns them off
bool afhip_decode_phases_supported(const afhip_llm_weights* w, int B);
// argmax partials, sums of squares, activation images
size_t afhip_decode_phases_scratch_bytes(const afhip_llm_weights* w, int B);
// the [8 or 16, n_q hd] fragment-order image the decode attention's merge writes and the o phase reads
void* afhip_decode_phases_att_image(const afhip_llm_weights* w, int B, void* scratch);

struct afhip_phase_step {            // one launch of a decode step
    const afhip_llm_weights* w;
    int B;
    int phase;                       // one AFHIP_PH_* value
    int layer;                       // decoder layer of the phase (q|k|v, o, gate/up, down)
    char* x; char* qkv;              // plain rows: residual stream [B, hidden], q|k|v [B, (n_q + 2 n_kv) hd]
    void* scratch;
    const afhip_decode_state* st;
    int step;
};
int afhip_decode_phase_launch(const afhip_phase_step& c, hipStream_t s);
