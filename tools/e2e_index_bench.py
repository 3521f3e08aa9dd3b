#!/usr/bin/env python3
"""Indexing throughput through the reference-shaped surfaces only: parsed files -> CodeChunker -> Embedder
(HipUniXcoderProvider, native tokenizer, dynamic batching) -> HipVectorStore.upsert, as VectorIndexer drives them
(src/lattice/embeddings/indexer.py:46-119).  Sequential index_files() as the reference does it, and the same files with
32 index_file() coroutines in flight (what a concurrent orchestrator gives the provider's batcher to coalesce).
A 12-layer checkpoint directory is synthesised.  python tools/e2e_index_bench.py [n_files]"""
import asyncio, glob, json, os, sys, tempfile, time, types
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import coderag_amd
from coderag_amd import encoder as drv
from safetensors.torch import save_file
from tokenizers import ByteLevelBPETokenizer

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 600
d = tempfile.mkdtemp()
srcs = sorted(glob.glob(os.path.join(ROOT, "**", "*.py"), recursive=True))
tr = ByteLevelBPETokenizer(add_prefix_space=False)
tr.train(srcs, vocab_size=8000, min_frequency=2, special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>", "<encoder-only>"])
tr.save_model(d)
cfg = drv.EncoderConfig(vocab_size=8000)
json.dump({"vocab_size": 8000, "hidden_size": 768, "num_hidden_layers": 12, "num_attention_heads": 12, "intermediate_size": 3072,
           "max_position_embeddings": 1026, "type_vocab_size": 10}, open(os.path.join(d, "config.json"), "w"))
save_file({k: torch.from_numpy(v) for k, v in drv.synthetic_weights(cfg, 31).items()}, os.path.join(d, "model.safetensors"))

text = "\n".join(open(f, encoding="utf-8", errors="ignore").read() for f in srcs)
lines = text.split("\n")
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


files = [parsed_file(i) for i in range(n_files)]


async def main():
    from coderag_amd.embedder import Embedder
    from coderag_amd.indexer import CodeChunker, VectorIndexer
    from coderag_amd.providers import HipUniXcoderProvider, ProviderConfig
    from coderag_amd.store import HipVectorStore
    for rows in ("list", "numpy"):
        provider = HipUniXcoderProvider(ProviderConfig(provider="unixcoder-hip", model=d, extra={"vector_rows": rows}))
        embedder = Embedder(provider_instance=provider)
        await embedder.embed_batch(["warm up"] * 64)
        print("vector_rows =", rows, flush=True)
        for mode in ("sequential", "32 in flight", "256 in flight"):
            async with HipVectorStore(dim=768, dtype="bf16", initial_capacity=1 << 16) as store:
                await store.create_collections()
                indexer = VectorIndexer(store, embedder, CodeChunker(max_tokens=1000, overlap_tokens=200))
                t0 = time.perf_counter()
                if mode == "sequential":
                    n = await indexer.index_files(files, project_name="proj")
                else:
                    sem = asyncio.Semaphore(int(mode.split()[0]))

                    async def one(f):
                        async with sem:
                            return await indexer.index_file(f, project_name="proj")
                    n = sum(await asyncio.gather(*(one(f) for f in files)))
                dt = time.perf_counter() - t0
                print(f"{mode}: {n} chunks of {len(files)} files in {dt:.2f} s = {n / dt:.0f} chunks/s", flush=True)

if os.environ.get("PROFILE") == "1":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    asyncio.run(main())
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
else:
    asyncio.run(main())
