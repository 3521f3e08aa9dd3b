#!/usr/bin/env python3
"""Texts -> embeddings through the driver's text surface (HipUniXcoder.embed_texts): where the time goes once the encoder itself
runs at ~25 k chunks/s.  A 12-layer checkpoint directory is synthesised (seeded weights, a vocabulary trained on this repo's
sources).  python tools/e2e_embed_bench.py [n_texts]"""
import glob, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coderag_amd
from coderag_amd import encoder as drv
from safetensors.torch import save_file
from tokenizers import ByteLevelBPETokenizer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
d = tempfile.mkdtemp()
files = sorted(glob.glob(os.path.join(ROOT, "**", "*.py"), recursive=True) + glob.glob(os.path.join(ROOT, "code-rag_amd", "csrc", "*")))
files = [f for f in files if "gpurun_out" not in f]
tr = ByteLevelBPETokenizer(add_prefix_space=False)
tr.train(files, vocab_size=8000, min_frequency=2, special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>", "<encoder-only>"])
tr.save_model(d)
cfg = drv.EncoderConfig(vocab_size=8000)
json.dump({"vocab_size": 8000, "hidden_size": 768, "num_hidden_layers": 12, "num_attention_heads": 12, "intermediate_size": 3072,
           "max_position_embeddings": 1026, "type_vocab_size": 10}, open(os.path.join(d, "config.json"), "w"))
save_file({k: torch.from_numpy(v) for k, v in drv.synthetic_weights(cfg, 31).items()}, os.path.join(d, "model.safetensors"))
texts = []
for f in files:
    s = open(f, encoding="utf-8", errors="ignore").read()
    texts += [s[i:i + 700] for i in range(0, len(s), 700)]
rng = np.random.default_rng(0)
texts = [texts[i] for i in rng.integers(0, len(texts), n)]
model = drv.load_unixcoder(d, device=0)
model.embed_texts(texts[:3000])
torch.cuda.synchronize()
t0 = time.perf_counter()
ids, lens = model.tok.encode_bodies(texts, 508)
t1 = time.perf_counter()
out = model.embed_bodies(ids, lens, max_tokens=65536)
t1b = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
arr = out.cpu().numpy()
t3 = time.perf_counter()
lst = arr.tolist()
t4 = time.perf_counter()
print(f"stages: tokenize {t1 - t0:.3f} s ({lens.sum() / (t1 - t0) / 1e6:.2f} M tok/s), pack+enqueue {t1b - t1:.3f} s, GPU tail {t2 - t1b:.3f} s, "
      f"D2H {t3 - t2:.3f} s, .tolist() {t4 - t3:.3f} s -> sequential {n / (t4 - t0):.0f} texts/s (mean {lens.mean():.0f} tokens)", flush=True)
for chunk in (1 << 30, 8192, 4096, 2048):
    model.PIPELINE_CHUNK = chunk
    for rows in ("list", "numpy"):
        t0 = time.perf_counter()
        res = model.embed_texts(texts, rows=rows)
        dt = time.perf_counter() - t0
        print(f"embed_texts chunk={chunk if chunk < 1 << 30 else 'off'} rows={rows}: {n / dt:.0f} texts/s ({dt:.3f} s)", flush=True)
