#!/usr/bin/env python3
"""Where one text query's time goes (1M rows): tokenizer, encoder driver, provider (asyncio + worker thread), index search
through the C ABI, store.search (asyncio + worker thread + payload projection), VectorSearcher.search_code end to end.
Median of 200 lone calls each.  python tools/query_path_breakdown.py [rows]"""
import asyncio, os, sys, time, uuid, tempfile, json, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coderag_amd
from coderag_amd import encoder as drv
from safetensors.torch import save_file
from tokenizers import ByteLevelBPETokenizer

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = tempfile.mkdtemp()
srcs = sorted(glob.glob(os.path.join(ROOT, "**", "*.py"), recursive=True))
tr = ByteLevelBPETokenizer(add_prefix_space=False)
tr.train(srcs, vocab_size=8000, min_frequency=2, special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>", "<encoder-only>"])
tr.save_model(d)
cfg = drv.EncoderConfig(vocab_size=8000)
json.dump({"vocab_size": 8000, "hidden_size": 768, "num_hidden_layers": 12, "num_attention_heads": 12, "intermediate_size": 3072,
           "max_position_embeddings": 1026, "type_vocab_size": 10}, open(os.path.join(d, "config.json"), "w"))
save_file({k: torch.from_numpy(v) for k, v in drv.synthetic_weights(cfg, 31).items()}, os.path.join(d, "model.safetensors"))
text = "how does the retry helper parse the configuration file and return a value"


def med(fn, n=200):
    for _ in range(10):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


async def amed(fn, n=200):
    for _ in range(10):
        await fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        await fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


async def main():
    from coderag_amd.embedder import Embedder
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    from coderag_amd.store import HipVectorStore
    from coderag_amd.vector_search import VectorSearcher
    provider = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model=d))
    embedder = Embedder(provider_instance=provider)
    await embedder.embed(text)
    model = drv.load_unixcoder(d, device=0)
    ids, lens = model.tok.encode_bodies([text], 508)
    print(f"tokenizer (native, 1 text)           {med(lambda: model.tok.encode_bodies([text], 508)):.3f} ms   ({int(lens[0])} tokens)")
    print(f"encoder driver embed_texts([text])   {med(lambda: model.embed_texts([text])):.3f} ms   (tokenize + H2D + forward + D2H + list)")
    print(f"provider.embed(text)                 {await amed(lambda: embedder.embed(text)):.3f} ms   (+ asyncio batching + worker thread)")
    rng = np.random.default_rng(0)
    async with HipVectorStore(dim=768, dtype="bf16", initial_capacity=rows) as s:
        await s.create_collections()
        for r0 in range(0, rows, 100_000):
            m = min(100_000, rows - r0)
            v = rng.standard_normal((m, 768)).astype(np.float32)
            await s.upsert("code_chunks", [str(uuid.UUID(int=r0 + i)) for i in range(m)], v,
                           [{"file_path": f"f{(r0 + i) % 5000}.py", "entity_name": f"e{r0 + i}", "language": "python", "content": "x" * 200,
                             "entity_type": "function", "start_line": 1, "end_line": 9} for i in range(m)])
        q = rng.standard_normal(768).astype(np.float32)
        idx = s._collections["code_chunks"].index
        print(f"index.search (C ABI, host in/out)    {med(lambda: idx.search(q[None, :], 10)):.3f} ms   ({rows} rows)")
        ql = q.tolist()
        print(f"store.search(list, limit=10)         {await amed(lambda: s.search('code_chunks', ql, limit=10)):.3f} ms   (+ asyncio + worker thread + hit dicts)")
        searcher = VectorSearcher(s, embedder)
        print(f"VectorSearcher.search_code(text)     {await amed(lambda: searcher.search_code(text, limit=10)):.3f} ms")

asyncio.run(main())
