"""UALM job pieces the inference path needs: unified vocabulary and the batch-dict builder.

Mirrors UALM/models/ualm/ualm_job.py: `_build_vocabulary` (:71-110), `UALMPreprocessor.preprocessing` (:311-418),
`collate_fn` bucket mode (:219-309) with `utils/data.py:pad_list` semantics, `find_length` (:200-217) and the task
templates of task_conf_ualm.py:18-31.  This is the caller side of the hot path: pure host logic,
it produces exactly the dict `ParallelLLM.inference(**batch)` consumes."""
import re
from typing import Dict, List

import numpy as np
import torch

UALM_TASK_CONFIGS = {
    "text_only": [("user", "text1"), ("assistant", "text2")],
    "caption_to_audio": [("user", "text1"), ("assistant", "audio1")],
    "audio_to_caption": [("user", "text1"), ("user", "audio1"), ("assistant", "text2")],
    "audio_to_conversation": [("user", "audio1"), ("user", "text1"), ("assistant", "text2")],
    "audio_only": [("user", "text1"), ("assistant", "audio1")],
    "transcription_to_speech": [("user", "text1"), ("assistant", "audio1")],
    "speech_to_transcription": [("user", "text1"), ("user", "audio1"), ("assistant", "text2")],
}

SPECIAL_TOKENS = ["<|pad|>", "<|bos|>", "<|eos|>", "<|eot|>", "<|system|>", "<|user|>", "<|assistant|>", "<|text|>",
                  "<|audio|>", "<|speech|>", "<|image|>", "<|video|>", "<|toolcall|>"]


def build_vocabulary(multimodal_io: dict, num_special_tokens: int = 256):
    """ualm_job.py:71-110."""
    vocab_intervals = {"special_token": [(0, num_special_tokens)]}
    vocab = list(SPECIAL_TOKENS)
    while len(vocab) < num_special_tokens:
        vocab.append(f"<|unused_{len(vocab)}|>")
    start = num_special_tokens
    for io_name, io in multimodal_io.items():
        if io.is_discrete:
            vocab.extend(io.get_vocabulary())
            vocab_intervals[io_name] = [(start + s, start + e) for s, e in io.get_stream_interval()]
            start = len(vocab)
    assert len(vocab) == len(set(vocab)), "There are duplicated tokens in the vocab"
    return vocab, vocab_intervals


def pad_list(sequences, pad_value: float = 0.0):
    """utils/data.py:16-90: right-pad along dim 0, stack; returns (padded, lengths)."""
    if not sequences:
        raise ValueError("Empty sequence list")
    ts = [torch.from_numpy(s) if isinstance(s, np.ndarray) else s for s in sequences]
    dtype = ts[0].dtype
    for t in ts[1:]:
        if t.shape[1:] != ts[0].shape[1:]:
            raise ValueError("All sequences must have the same shape except for the first dimension")
        dtype = torch.promote_types(dtype, t.dtype)
    lens = [t.shape[0] for t in ts]
    out = torch.full((len(ts), max(lens)) + tuple(ts[0].shape[1:]), pad_value, dtype=dtype)
    for i, t in enumerate(ts):
        out[i, : lens[i]] = t.to(dtype)
    return out, torch.tensor(lens, dtype=torch.long)


class UALMPreprocessor:
    def __init__(self, is_train, multimodal_io, vocab, vocab_intervals, audio_input: str = "continuous_audio",
                 audio_output: str = "discrete_audio", loss_region: str = "assistant", batchfy_method: str = "bucket",
                 audio_cfg: float = 0.0):
        if is_train:
            raise NotImplementedError("training-side preprocessing (CFG dropout, packing) is out of scope")
        self.is_train = is_train
        self.multimodal_io = multimodal_io
        self.audio_input, self.audio_output, self.loss_region = audio_input, audio_output, loss_region
        self.batchfy_method = batchfy_method
        self.vocab, self.vocab_intervals = vocab, vocab_intervals
        self.pad_id = vocab.index("<|pad|>")
        streams = [io.num_stream() for io in multimodal_io.values() if io.is_discrete]
        if not streams:
            raise ValueError("You should have at least one discrete multimodal IO")
        self.num_stream = max(streams)

    def special_token(self, token):
        r = np.ones((1, self.num_stream)).astype(np.int64) * self.pad_id
        r[0, 0] = self.vocab[: self.vocab_intervals["special_token"][0][1]].index(token)
        return r

    def special_mask(self, value):
        r = np.zeros((1, self.num_stream)).astype(np.float32)
        r[0, 0] = value
        return r

    @staticmethod
    def _reformat_data_dict(data_dict):
        out, ai = {}, 1
        for k, v in data_dict.items():
            if k == "audio":
                out[f"audio{ai}"] = v
                ai += 1
        ti = 1
        for row in data_dict["text"]:
            if row[1] == "text":
                out[f"text{ti}"] = row[2]
                ti += 1
        return out

    def _apply_chat_template(self, task, data_dict):
        if "dialogue" in data_dict:
            if len(data_dict) != 1:
                raise ValueError("If dialogue exist, there should be no more other entries")
            assert all(m[0] != "assistant" for m in data_dict["dialogue"]), "during inference, input dialogue should not contain model output (assistant message)"
            return data_dict["dialogue"]
        data_dict = self._reformat_data_dict(data_dict)
        msgs = []
        for role, entry in UALM_TASK_CONFIGS[task]:
            if role == "assistant":
                break
            if re.match(r"^audio", entry):
                io = self.audio_input if role in ("user", "system") else self.audio_output
            elif re.match(r"^text", entry):
                io = "text"
            else:
                raise ValueError(f"Not supported data entry in template: {entry}")
            msgs.append((role, io, data_dict[entry]))
        return msgs

    def find_length(self, key, data_dict):
        length = 1
        for _, io, data in self._apply_chat_template(key[0], data_dict):
            length += 3 + self.multimodal_io[io].find_length(data)
        return length

    def preprocessing(self, key, data_dict):
        task = key[0]
        messages = self._apply_chat_template(task, data_dict)
        seq, conti, masks = [self.special_token("<|bos|>")], [], [self.special_mask(0.0)]
        accum = 1
        eots = [a[0] == b[0] for a, b in zip(messages[:-1], messages[1:])] + [False]
        for apply_eot, (role, this_io, this_data) in zip(eots, messages):
            apply_loss = float(role == "assistant" or self.loss_region == "all")
            sm = self.special_mask(apply_loss)
            seq.append(self.special_token(f"<|{role}|>"))
            masks.append(sm)
            modality = self.multimodal_io[this_io].modality
            if modality == "audio":
                if task in ["caption_to_audio", "audio_to_caption", "audio_to_conversation", "audio_only"]:
                    seq.append(self.special_token("<|audio|>"))
                elif task in ["transcription_to_speech", "speech_to_transcription"]:
                    seq.append(self.special_token("<|speech|>"))
                else:
                    seq.append(self.special_token(f"<|{modality}|>"))
            else:
                seq.append(self.special_token(f"<|{modality}|>"))
            masks.append(sm)
            accum += 2
            this_seq, conti_feat, loss_mask = self.multimodal_io[this_io].preprocess(this_data)
            assert this_seq.shape == loss_mask.shape
            if self.multimodal_io[this_io].is_discrete:
                bias = self.vocab_intervals[this_io][0][0]
                this_seq = np.where(this_seq == self.pad_id, self.pad_id, this_seq + bias)
            if this_seq.shape[1] < self.num_stream:
                this_seq = np.pad(this_seq, ((0, 0), (0, self.num_stream - this_seq.shape[1])))
            seq.append(this_seq)
            if conti_feat is not None:
                conti.append((this_io, accum, conti_feat[0], conti_feat[1]))
            if loss_mask.shape[1] < self.num_stream:
                loss_mask = np.pad(loss_mask, ((0, 0), (0, self.num_stream - loss_mask.shape[1])))
            masks.append(loss_mask * apply_loss)
            accum += this_seq.shape[0]
            seq.append(self.special_token("<|eot|>" if apply_eot else "<|eos|>"))
            masks.append(sm)
            accum += 1
        return {"sequence": np.concatenate(seq, axis=0), "conti_feats": conti, "loss_mask": np.concatenate(masks, axis=0)}

    def collate_fn(self, data_lst):
        if self.batchfy_method != "bucket":
            raise NotImplementedError("only bucket batching is built (pack is a training-side method)")
        ret = {"keys": []}
        dicts = []
        for key, dd in data_lst:
            try:
                dicts.append(self.preprocessing(key, dd))
                ret["keys"].append(key)
            except Exception as e:   # ualm_job.py:236-250: bad samples are dropped, not fatal
                print(f"Error <{e}> processing sample <{key}>")
        if not dicts:
            raise ValueError("No valid samples after preprocessing")
        seqs, _ = pad_list([torch.from_numpy(d["sequence"]) for d in dicts])
        lms, _ = pad_list([torch.from_numpy(d["loss_mask"]) for d in dicts])
        ret["seqs"], ret["loss_masks"] = seqs, lms
        by_io = {}
        for b, d in enumerate(dicts):
            for io, start, length, feat in d["conti_feats"]:
                by_io.setdefault(io, [[], []])
                by_io[io][0].append((b, start, length))
                by_io[io][1].append(feat)
        for io, (idx, feats) in by_io.items():
            ret[f"{io}_indices"] = torch.Tensor(idx).long()
            ret[f"{io}_feats"], ret[f"{io}_lengths"] = pad_list(feats)
        return ret
