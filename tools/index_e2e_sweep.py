#!/usr/bin/env python3
"""index_files_batched through the reference-shaped surfaces at several pipeline group sizes, and the stages on their own
(chunking, one embed_array call over every chunk, one upsert): where the time of bench.py's index_e2e leg goes.
python tools/index_e2e_sweep.py"""
import asyncio, os, sys, time, types
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coderag_amd
import bench
from coderag_amd.embedder import Embedder
from coderag_amd.indexer import CodeChunker, VectorIndexer
from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
from coderag_amd.store import HipVectorStore

box = {}
d = bench.synth_checkpoint(np, torch, box)
lines = "\n".join(open(f, encoding="utf-8", errors="ignore").read() for f in bench._source_files() if f.endswith(".py")).split("\n")
rng = np.random.default_rng(0)


def parsed_file(i):
    ents = []
    for j in range(int(rng.integers(8, 40))):
        a = int(rng.integers(0, len(lines) - 40))
        code = "\n".join(lines[a:a + int(rng.integers(6, 40))])
        ents.append(types.SimpleNamespace(type=types.SimpleNamespace(value="function"), name=f"fn_{i}_{j}", qualified_name=f"mod{i}.fn_{i}_{j}",
                                          signature=f"def fn_{i}_{j}(x)", docstring="Does things.", code=code, start_line=10 * j + 1, end_line=10 * j + 9))
    info = types.SimpleNamespace(path=Path(f"/proj/mod{i}.py"), content_hash=f"h{i}", language=types.SimpleNamespace(value="python"))
    return types.SimpleNamespace(file_info=info, content="", all_entities=ents)


files = [parsed_file(i) for i in range(600)]
provider = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model=d))
embedder = Embedder(provider_instance=provider)
chunker = CodeChunker(max_tokens=1000, overlap_tokens=200)


async def stages():
    await embedder.embed_batch(["warm up"] * 64)
    t0 = time.perf_counter()
    chunks = [c for f in files for c in chunker.chunk_file(f, project_name="proj")]
    t_chunk = time.perf_counter() - t0
    texts = [c.content for c in chunks]
    await embedder.embed_array(texts[:4096])
    t0 = time.perf_counter()
    vecs = await embedder.embed_array(texts)
    t_embed = time.perf_counter() - t0
    model = provider._load()
    ids, lens = model.tok.encode_bodies(texts, 508)
    t0 = time.perf_counter()
    model.tok.encode_bodies(texts, 508)
    t_tok = time.perf_counter() - t0
    import uuid
    t0 = time.perf_counter()
    ids_ = [str(uuid.uuid4()) for _ in chunks]
    pay = [c.to_payload() for c in chunks]
    t_tab = time.perf_counter() - t0
    async with HipVectorStore(dim=768, dtype="bf16", initial_capacity=1 << 16) as store:
        await store.create_collections()
        t0 = time.perf_counter()
        await store.upsert("code_chunks", ids_, vecs, pay)
        t_up = time.perf_counter() - t0
    print(f"{len(chunks)} chunks, mean {np.minimum(lens, 508).mean() + 4:.0f} tokens: chunking {t_chunk:.3f} s, tokenizer alone {t_tok:.3f} s, embed_array (one call) {t_embed:.3f} s "
          f"= {len(chunks) / t_embed:.0f} chunks/s, ids + payload dicts {t_tab:.3f} s, upsert {t_up:.3f} s", flush=True)


async def sweep():
    for group, first in ((4096, 512), (4096, 512), (8192, 512), (4096, 1024), (2048, 512), (8192, 8192), (1 << 30, 1 << 30)):
        VectorIndexer.GROUP, VectorIndexer.GROUP_FIRST = group, first
        async with HipVectorStore(dim=768, dtype="bf16", initial_capacity=1 << 16) as store:
            await store.create_collections()
            indexer = VectorIndexer(store, embedder, chunker)
            t0 = time.perf_counter()
            n = await indexer.index_files_batched(files, project_name="proj")
            dt = time.perf_counter() - t0
            print(f"GROUP {group} first {first}: {n} chunks in {dt:.3f} s = {n / dt:.0f} chunks/s", flush=True)

asyncio.run(stages())
asyncio.run(sweep())
