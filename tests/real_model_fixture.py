"""A complete local model directory, built from nothing (no download): the only thing the harness
ever serves is such a directory (/root/reference/benchmark-mlflow/vllm_processor.py:29 mounts
`/data/local_models/<name>`; /root/reference/benchmark-eval/main.py:105 polls the server started on it).

    model-00001-of-0000N.safetensors ...   the seeded tiny checkpoint of oracle/tiny_models.py, bf16, split
    model.safetensors.index.json           over N shard files with the HF weight map
    config.json, generation_config.json    the latter may carry a LIST of end-of-sequence ids
    tokenizer.json                         a byte-level BPE trained in memory with `tokenizers`, with a
                                           post-processor that prepends BOS (as a Llama-3 tokenizer's does)
    tokenizer_config.json                  bos/eos token strings and a chat template that emits bos_token
"""
import json
import os

import numpy as np

BOS, EOS, EOT = "<|bos|>", "<|eos|>", "<|eot|>"

CORPUS = [
    "the quick brown fox jumps over the lazy dog", "a benchmark harness launches the engine and waits",
    "hello world, hello there, hello again", "decode tokens per second and time to first token",
    "paged attention reads the key and value pages of every sequence", "what is the capital of France?",
    "the answer is forty two", "user assistant system", "stop here and go on", "tokens tokens tokens and more tokens",
    "0 1 2 3 4 5 6 7 8 9 10 11 12", "naive café résumé über straße 日本語",
]

# Llama-3 style: begins with bos_token, one header per message, ends where the assistant's answer starts
CHAT_TEMPLATE = ("{{ bos_token }}{% for m in messages %}<|{{ m['role'] }}|>\n{{ m['content'] }}" + EOT + "\n{% endfor %}"
                 "{% if add_generation_prompt %}<|assistant|>\n{% endif %}")


def train_tokenizer(vocab_size: int):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    tk = Tokenizer(models.BPE())
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=vocab_size, special_tokens=[BOS, EOS, EOT],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tk.train_from_iterator(CORPUS * 4, trainer)
    tk.post_processor = processors.TemplateProcessing(single=f"{BOS} $A", pair=f"{BOS} $A $B",
                                                      special_tokens=[(BOS, tk.token_to_id(BOS))])
    return tk


def build_model_dir(path, name="llama_d128", shards=2, eos_config=None, eos_generation=None, chat_template=CHAT_TEMPLATE,
                    strip_model_prefix=False):
    """Writes the directory and returns (config dict incl. architectures, weights {name: fp32 array on the bf16 grid},
    tokenizers.Tokenizer).  `eos_config` -> config.json eos_token_id (default: the tokenizer's EOS id);
    `eos_generation` -> generation_config.json eos_token_id (int or list; the file is left out when None)."""
    import torch
    from safetensors.torch import save_file

    from oracle.tiny_models import make_weights

    os.makedirs(path, exist_ok=True)
    arch, cfg, w = make_weights(name)
    tk = train_tokenizer(min(cfg["vocab_size"], 480))
    assert tk.get_vocab_size() <= cfg["vocab_size"]
    tk.save(os.path.join(path, "tokenizer.json"))
    eos_id = tk.token_to_id(EOS)
    full = {**cfg, "architectures": [arch], "bos_token_id": tk.token_to_id(BOS),
            "eos_token_id": eos_id if eos_config is None else eos_config, "torch_dtype": "bfloat16"}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(full, f)
    if eos_generation is not None:
        with open(os.path.join(path, "generation_config.json"), "w") as f:
            json.dump({"bos_token_id": tk.token_to_id(BOS), "eos_token_id": eos_generation, "do_sample": False}, f)
    meta = {"bos_token": BOS, "eos_token": {"content": EOS, "lstrip": False, "rstrip": False},
            "tokenizer_class": "PreTrainedTokenizerFast"}
    if chat_template is not None:
        meta["chat_template"] = chat_template
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump(meta, f)

    names = list(w)
    per = (len(names) + shards - 1) // shards
    weight_map, total = {}, 0
    for s in range(shards):
        fname = f"model-{s + 1:05d}-of-{shards:05d}.safetensors" if shards > 1 else "model.safetensors"
        part = {}
        for n in names[s * per:(s + 1) * per]:
            key = n[len("model."):] if strip_model_prefix and n.startswith("model.") else n
            part[key] = torch.from_numpy(np.ascontiguousarray(w[n])).to(torch.bfloat16)
            weight_map[key] = fname
            total += part[key].numel() * 2
        save_file(part, os.path.join(path, fname), metadata={"format": "pt"})
    if shards > 1:
        with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f)
    return full, w, tk
