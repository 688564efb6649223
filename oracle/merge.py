"""CPU restatement (numpy) of the AF3 / Qwen2-Audio placeholder expansion
`Qwen2AudioForConditionalGeneration._merge_input_ids_with_audio_features`
(/root/reference/UALM/models/ualm/multimodal_io/modeling_whisper.py:913-1108).

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): the product path is
audio_intelligence_amd.multimodal_io.modeling_whisper.merge_input_ids_with_audio_features (HIP row gather).
Pinned by tests/golden/golden_merge.npz, captured from the reference function itself (oracle/make_golden_merge.py).
"""
import numpy as np


def merge_plan(num_audio_tokens, input_ids, attention_mask, audio_token_index, padding_side="left"):
    """Index plan of the merge (everything but the embedding rows), following the reference step by step.

    Returns dict(max_len, left_padding, text_b, text_src, text_dst, audio_dst (flat positions in row-major [B, max_len]
    order, one per valid audio row in stacking order)).
    """
    input_ids = np.asarray(input_ids, np.int64)
    attention_mask = np.asarray(attention_mask, np.int64)
    num_audio_tokens = np.asarray(num_audio_tokens, np.int64)
    B, L = input_ids.shape
    # :1003-1019 padding side: inferred from the mask, `padding_side` only when both edge columns are all ones
    lp, rp = bool(np.any(attention_mask[:, 0] == 0)), bool(np.any(attention_mask[:, -1] == 0))
    left_padding = True
    if B > 1:
        if lp and not rp:
            left_padding = True
        elif not lp and rp:
            left_padding = False
        elif not lp and not rp:
            left_padding = padding_side == "left"
        else:
            raise ValueError(f"both side of attention_mask has zero, invalid. {attention_mask}")
    special = input_ids == audio_token_index                               # :1022
    text_b, text_src = np.where((input_ids != audio_token_index) & (attention_mask == 1))   # :1031-1033
    # :1039-1046 every <|AUDIO|> token widens to num_audio_tokens of "its" audio, in row-major order of appearance
    width = np.zeros_like(input_ids)
    width[special] = num_audio_tokens - 1                                  # raises like the reference if counts differ
    width = width + 1
    new_pos = np.cumsum(width, -1) - 1
    max_len = int(width.sum(-1).max())
    nb_pad = max_len - 1 - new_pos[:, -1]
    if left_padding:
        new_pos = new_pos + nb_pad[:, None]
    text_dst = new_pos[text_b, text_src]
    # :1078-1096 audio slots = positions not written by text, minus the padded side
    audio_slot = np.ones((B, max_len), bool)
    audio_slot[text_b, text_dst] = False
    seq = np.arange(max_len)[None, :]
    valid = width.sum(-1) - (attention_mask == 0).sum(-1)
    if left_padding:
        val = (max_len - seq) <= valid[:, None]
    else:
        val = seq < valid[:, None]
    audio_slot &= val
    if int(audio_slot.sum()) != int(num_audio_tokens.sum()):               # :1098-1102
        raise ValueError("The input provided to the model are wrong. The number of audio tokens is "
                         f"{special.sum(-1)} while the number of audio given to the model is {len(num_audio_tokens)}. "
                         "This prevents correct indexing and breaks batch generation.")
    return dict(max_len=max_len, left_padding=left_padding, text_b=text_b, text_src=text_src, text_dst=text_dst,
                audio_dst=np.flatnonzero(audio_slot.reshape(-1)), audio_slot=audio_slot)


def merge_input_ids_with_audio_features(audio_features, num_audio_tokens, inputs_embeds, input_ids, attention_mask, labels,
                                        audio_token_index, pad_token_id=-1, ignore_index=-100, padding_side="left"):
    """-> (final_embedding, final_attention_mask, final_labels | None, position_ids, final_input_ids), reference :1056-1108."""
    audio_features = np.asarray(audio_features)
    inputs_embeds = np.asarray(inputs_embeds)
    input_ids = np.asarray(input_ids, np.int64)
    attention_mask = np.asarray(attention_mask, np.int64)
    num_audio_tokens = np.asarray(num_audio_tokens, np.int64)
    n_audio, max_tok, H = audio_features.shape
    B, L = input_ids.shape
    plan = merge_plan(num_audio_tokens, input_ids, attention_mask, audio_token_index, padding_side)
    M = plan["max_len"]
    emb = np.zeros((B, M, H), inputs_embeds.dtype)
    mask = np.zeros((B, M), attention_mask.dtype)
    ids = np.full((B, M), pad_token_id, input_ids.dtype)
    tb, ts, td = plan["text_b"], plan["text_src"], plan["text_dst"]
    emb[tb, td] = inputs_embeds[tb, ts]
    mask[tb, td] = attention_mask[tb, ts]
    ids[tb, td] = input_ids[tb, ts]
    final_labels = None
    if labels is not None:
        final_labels = np.full((B, M), ignore_index, np.int64)
        final_labels[tb, td] = np.asarray(labels, np.int64)[tb, ts]
    keep = np.arange(max_tok)[None, :] < num_audio_tokens[:, None]          # :999-1002 valid rows of every audio, stacked
    emb.reshape(B * M, H)[plan["audio_dst"]] = audio_features[keep]
    mask |= plan["audio_slot"].astype(mask.dtype)
    pos = np.cumsum(mask, -1) - 1
    pos[mask == 0] = 1                                                      # :1106
    return emb, mask, final_labels, pos, ids
