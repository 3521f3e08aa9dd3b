#!/usr/bin/env python3
"""Generate tests/golden/*reference*.json by RUNNING the reference's own hot-path files.

Survey-container-only tool: needs /root/reference (read-only; absent on the GPU box) and is never
imported by the package or the tests.  It loads individual reference files by path
(importlib.util.spec_from_file_location) because `import lattice` fails on absent third-party
packages; the absent packages are given inert placeholders in sys.modules so
the module-level `import` statements succeed -- none of their behaviour is exercised except where noted.
What is written out is DATA only: inputs fed to the reference and the outputs it returned.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_goldens.py
"""
import asyncio
import dataclasses
import importlib.util
import json
import os
import sys
import types
from pathlib import Path
from unittest.mock import AsyncMock, MagicMock

sys.dont_write_bytecode = True
REF = Path("/root/reference/src/lattice")
OUT = Path(__file__).resolve().parent


def shell(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, REF / rel)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install_placeholders():
    def retry(*a, **k):
        return lambda fn: fn
    shell("tenacity", retry=retry, stop_after_attempt=lambda *a, **k: None, wait_exponential=lambda *a, **k: None)
    shell("openai", AsyncOpenAI=type("AsyncOpenAI", (), {"__init__": lambda self, *a, **k: None}))
    shell("tiktoken", get_encoding=lambda name: types.SimpleNamespace(encode=lambda text: text.split()))

    class _Any:
        def __init__(self, *a, **k):
            self.__dict__.update(k)

        def __class_getitem__(cls, item):
            return cls
    models = types.SimpleNamespace(**{n: _Any for n in ("Filter", "FieldCondition", "MatchValue", "PointStruct", "VectorParams",
                                                         "FilterSelector", "CollectionInfo")},
                                   Distance=types.SimpleNamespace(COSINE="Cosine"),
                                   PayloadSchemaType=types.SimpleNamespace(KEYWORD="keyword"))
    shell("qdrant_client", AsyncQdrantClient=_Any, models=models)
    settings = types.SimpleNamespace(llm_model="gpt-4o", openai_api_key="", max_concurrent_requests=5, qdrant_host="localhost",
                                     qdrant_port=6333, qdrant_grpc_port=6334, embedding_dimensions=768, chunk_max_tokens=1000,
                                     chunk_overlap_tokens=200)
    shell("lattice")
    shell("lattice.config", get_settings=lambda: settings)
    shell("lattice.core")
    shell("lattice.providers")
    shell("lattice.query")
    shell("lattice.query.graph_reasoning")
    shell("lattice.query.ranking")
    shell("lattice.parsing")
    shell("lattice.embeddings")


def load_reference():
    install_placeholders()
    R = types.SimpleNamespace()
    R.types = load("lattice.core.types", "core/types.py")
    R.errors = load("lattice.core.errors", "core/errors.py")
    R.base = load("lattice.providers.base", "providers/base.py")
    R.planner = load("lattice.query.query_planner", "query/query_planner.py")
    R.gmodels = load("lattice.query.graph_reasoning.models", "query/graph_reasoning/models.py")
    sys.modules["lattice.query.graph_reasoning"].GraphContext = R.gmodels.GraphContext
    sys.modules["lattice.query.graph_reasoning"].GraphNode = R.gmodels.GraphNode
    R.rmodels = load("lattice.query.ranking.models", "query/ranking/models.py")
    R.scorer = load("lattice.query.ranking.scorer", "query/ranking/scorer.py")
    R.ranker = load("lattice.query.ranking.ranker", "query/ranking/ranker.py")
    R.rutils = load("lattice.query.ranking.utils", "query/ranking/utils.py")
    R.pmodels = load("lattice.parsing.models", "parsing/models.py")
    R.chunker = load("lattice.embeddings.chunker", "embeddings/chunker.py")
    R.client = load("lattice.embeddings.client", "embeddings/client.py")
    sys.modules["lattice.providers"].get_embedding_provider = lambda **kw: None
    R.embedder = load("lattice.embeddings.embedder", "embeddings/embedder.py")
    R.indexer = load("lattice.embeddings.indexer", "embeddings/indexer.py")
    R.vsearch = load("lattice.query.vector_search", "query/vector_search.py")
    return R


# ---------------------------------------------------------------------------------------------- ranking
def node(name, file="src/a.py", ntype="function", qn=None, line=1, depth=None, **kw):
    d = dict(node_type=ntype, name=name, qualified_name=qn or name, file_path=file, start_line=line, end_line=line + 5,
             signature=kw.get("signature"), docstring=kw.get("docstring"), summary=kw.get("summary"),
             metadata=({"depth": depth} if depth is not None else {}))
    return d


def vhit(score, name, file="src/a.py", line=1, content_len=150, gid=None, summary=None, etype="function"):
    return dict(score=score, file_path=file, entity_type=etype, entity_name=name, language="python",
                content=("x" * content_len) if content_len else None, start_line=line, end_line=line + 5, graph_node_id=gid,
                summary=summary)


def ranking_scenarios(R):
    intents = [i.value for i in R.planner.QueryIntent]
    base_graph = {
        "primary_entities": [node("UserRepository", ntype="class", qn="app.repo.UserRepository", line=10, summary="Stores users",
                                  docstring="Repo", signature="class UserRepository")],
        "callers": [node("login", file="src/auth.py", line=20, depth=1, signature="def login()"),
                    node("refresh", file="src/auth.py", line=60, depth=3), node("deep", file="src/x.py", line=5, depth=6)],
        "callees": [node("hash_password", file="src/crypto.py", line=7, depth=2, docstring="hash")],
        "methods": [node("verify_password", qn="app.repo.UserRepository.verify_password", line=30, summary="checks")],
        "parent_classes": [node("BaseRepository", ntype="class", file="src/base.py", line=3)],
        "child_classes": [node("CachedUserRepository", ntype="class", file="src/cache.py", line=9)],
    }
    base_vec = [vhit(0.91, "verify_password", line=30, content_len=150, gid="app.repo.UserRepository.verify_password"),
                vhit(0.83, "UserRepository.save", file="src/a.py", line=80, content_len=2500),
                vhit(0.62, "unrelated", file="src/z.py", line=2, content_len=10),
                vhit(0.55, "nocontent", file="src/z.py", line=40, content_len=0),
                vhit(0.51, "huge", file="src/big.py", line=1, content_len=4000)]
    cent = {"app.repo.UserRepository": {"in_degree": 30, "out_degree": 45, "total_degree": 75, "relationship_count": 75},
            "app.repo.UserRepository.verify_password": {"in_degree": 4, "out_degree": 6, "total_degree": 10, "relationship_count": 10},
            "login": {"in_degree": 1, "out_degree": 1, "total_degree": 2, "relationship_count": 2}}
    ents = ["UserRepository", "verify_password"]
    sc = []
    for it in intents:
        sc.append(dict(name=f"intent_{it}", intent=it, entities=ents, graph=base_graph, vector=base_vec, centrality=cent))
    sc.append(dict(name="survey_example", intent="explain_implementation", entities=ents, graph={},
                   vector=[vhit(0.9, "verify_password", content_len=150), vhit(0.8, "other", file="src/b.py", content_len=10)],
                   centrality={"verify_password": {"total_degree": 10}}))
    sc.append(dict(name="merge_graph_and_vector", intent="locate_entity", entities=["parse_config"],
                   graph={"primary_entities": [node("parse_config", file="src/cfg.py", line=12, signature="def parse_config(p)")],
                          "callees": [node("read_file", file="src/io.py", line=3, depth=1)]},
                   vector=[vhit(0.77, "parse_config", file="src/cfg.py", line=12, content_len=300, summary="parses"),
                           vhit(0.7, "read_file", file="src/io.py", line=3, content_len=120),
                           vhit(0.7, "read_file", file="src/io.py", line=3, content_len=900)],
                   centrality={"parse_config": {"total_degree": 50}}))
    sc.append(dict(name="per_file_cap", intent="search_functionality", entities=[], graph={},
                   vector=[vhit(0.9 - 0.01 * i, f"f{i}", file="src/one.py", line=10 * i) for i in range(9)]
                   + [vhit(0.5, "g", file="src/two.py", line=1)], centrality=None))
    sc.append(dict(name="total_cap", intent="find_similar", entities=[], graph={},
                   vector=[vhit(0.99 - 0.005 * i, f"h{i}", file=f"src/m{i // 2}.py", line=i) for i in range(70)], centrality={}))
    sc.append(dict(name="empty_entity_name_quirk", intent="search_pattern", entities=[""], graph={},
                   vector=[vhit(0.6, "anything", line=3), vhit(0.6, "else", file="src/q.py", line=4)], centrality={}))
    sc.append(dict(name="ties_keep_store_order", intent="find_similar", entities=[], graph={},
                   vector=[vhit(0.5, f"t{i}", file=f"src/t{i}.py", line=1, content_len=500) for i in range(6)], centrality={}))
    return sc


def run_ranking(R, s):
    P, G = R.planner, R.gmodels
    plan = P.QueryPlan(original_query=s["name"], primary_intent=P.QueryIntent(s["intent"]), sub_queries=[],
                       entities=[P.ExtractedEntity(name=e) for e in s["entities"]], relationships=[])
    roles = {k: [G.GraphNode(**n) for n in s["graph"].get(k, [])]
             for k in ("primary_entities", "callers", "callees", "parent_classes", "child_classes", "methods")}
    ctx = G.GraphContext(containing_class=None, file_context=[], dependencies=[], dependents=[], call_chains=[],
                         inheritance_chains=[], **roles)
    ranked = R.ranker.HybridRanker().rank_results(plan, ctx, [dict(v) for v in s["vector"]], s["centrality"])
    return {"ranked": [dataclasses.asdict(r) for r in ranked],
            "flattened": R.rutils.ranked_results_to_search_results(ranked)}


# ---------------------------------------------------------------------------------------------- call shapes
def call_log(mock):
    return [[name, list(args), {k: v for k, v in kwargs.items()}] for name, args, kwargs in mock.mock_calls]


async def run_callshapes(R):
    out = {}

    # a-5: BaseEmbeddingProvider.embed_batch slicing
    sizes = []

    class Rec(R.base.BaseEmbeddingProvider):
        async def _embed_impl(self, texts):
            sizes.append(len(texts))
            return [[float(t)] for t in texts]
    p = Rec(R.base.ProviderConfig(provider="rec", model="m"))
    res = await p.embed_batch([str(i) for i in range(250)], batch_size=100)
    out["a5_embed_batch"] = {"impl_call_sizes": list(sizes), "n_out": len(res), "order_ok": [r[0] for r in res] == [float(i) for i in range(250)]}
    one = await p.embed("7")
    out["a5_embed_single"] = {"impl_call_sizes_after": sizes[-1], "value": one}

    # a-7: Embedder.embed_with_progress
    a7 = {}
    for n in (0, 1, 100, 101, 250):
        prov = MagicMock()
        prov.config = types.SimpleNamespace(provider="p", model="m")
        calls = []

        async def eb(batch, batch_size, _calls=calls):
            _calls.append([len(batch), batch_size])
            return [[0.0]] * len(batch)
        prov.embed_batch = eb
        sys.modules["lattice.providers"].get_embedding_provider = lambda **kw: prov
        R.embedder.get_embedding_provider = lambda **kw: prov
        e = R.embedder.Embedder()
        prog = []
        res = await e.embed_with_progress(["t"] * n, progress_callback=lambda d, t: prog.append([d, t]))
        a7[str(n)] = {"provider_calls": calls, "progress": prog, "n_out": len(res)}
    out["a7_embed_with_progress"] = a7

    # shared fixtures for a-9 / a-11 / a-12
    PM, T = R.pmodels, R.types
    fi = PM.FileInfo(path=Path("/project/main.py"), relative_path="main.py", language=T.Language.PYTHON, content_hash="hash123",
                     size_bytes=100, line_count=10)
    pf = PM.ParsedFile(file_info=fi, content="def hello(): pass", imports=[], entities=[
        PM.CodeEntity(type=T.EntityType.FUNCTION, name="hello", qualified_name="hello", signature="def hello()",
                      code="def hello(): pass", start_line=1, end_line=2)])
    CC = R.chunker.CodeChunk
    two = [CC(content="def hello(): pass", file_path="/project/main.py", entity_type="function", entity_name="hello", language="python",
              start_line=1, end_line=2),
           CC(content="def world(): pass", file_path="/project/main.py", entity_type="function", entity_name="world", language="python",
              start_line=4, end_line=5, graph_node_id="world", content_hash="hash123", project_name="proj")]

    def mocks(needs_update=True):
        q = AsyncMock()
        q.file_needs_update = AsyncMock(return_value=needs_update)
        e = AsyncMock()
        e.embed = AsyncMock(return_value=[0.1] * 4)
        e.embed_with_progress = AsyncMock(return_value=[[0.1] * 4, [0.2] * 4])
        c = MagicMock()
        c.chunk_file = MagicMock(return_value=two)
        return q, e, c

    def scrub(log):  # uuid4 ids are random: keep their count and type only
        for entry in log:
            kw = entry[2]
            if "ids" in kw:
                kw["ids"] = {"n": len(kw["ids"]), "all_uuid4_str": all(isinstance(i, str) and len(i) == 36 for i in kw["ids"])}
            if "progress_callback" in kw:
                kw["progress_callback"] = None if kw["progress_callback"] is None else "callable"
            entry[1] = [("<ParsedFile>" if isinstance(a, PM.ParsedFile) else a) for a in entry[1]]
        return log
    a9 = {}
    q, e, c = mocks()
    n = await R.indexer.VectorIndexer(q, e, c).index_file(pf, project_name="my-project")
    a9["index_file"] = {"returned": n, "store": scrub(call_log(q)), "embedder": scrub(call_log(e)), "chunker": scrub(call_log(c))}
    q, e, c = mocks(needs_update=False)
    n = await R.indexer.VectorIndexer(q, e, c).index_file(pf)
    a9["skip_unchanged"] = {"returned": n, "store": scrub(call_log(q)), "embedder": scrub(call_log(e))}
    q, e, c = mocks(needs_update=False)
    n = await R.indexer.VectorIndexer(q, e, c).index_file(pf, force=True)
    a9["force"] = {"returned": n, "store": scrub(call_log(q))}
    q, e, c = mocks()
    c.chunk_file = MagicMock(return_value=[])
    n = await R.indexer.VectorIndexer(q, e, c).index_file(pf)
    a9["no_chunks"] = {"returned": n, "store": scrub(call_log(q)), "embedder": scrub(call_log(e))}
    q, e, c = mocks()
    n = await R.indexer.VectorIndexer(q, e, c).index_files([pf, pf], project_name="p")
    a9["index_files"] = {"returned": n, "upserts": q.upsert.call_count}
    q, e, c = mocks()
    await R.indexer.VectorIndexer(q, e, c).index_summary(file_path="/project/main.py", entity_type="function", entity_name="hello",
                                                         summary="This function says hello", graph_node_id="hello")
    a9["index_summary"] = {"store": scrub(call_log(q)), "embedder": scrub(call_log(e))}
    q, e, c = mocks()
    e.embed_with_progress.side_effect = Exception("API Error")
    try:
        await R.indexer.VectorIndexer(q, e, c).index_file(pf)
    except R.errors.IndexingError as ex:
        a9["error"] = {"type": type(ex).__name__, "stage": ex.stage, "str": str(ex)}
    q, e, c = mocks()
    e.embed_with_progress.side_effect = Exception("API Error")
    n = await R.indexer.VectorIndexer(q, e, c).index_files([pf, pf])
    a9["index_files_swallows"] = {"returned": n}
    out["a9_vector_indexer"] = a9
    out["a8_payload"] = two[1].to_payload()

    hits = [{"id": "1", "score": 0.95, "payload": {"file_path": "/project/main.py", "entity_type": "function", "entity_name": "hello",
                                                    "content": "def hello(): pass", "start_line": 1, "end_line": 2, "language": "python",
                                                    "graph_node_id": "hello"}},
            {"id": "2", "score": 0.85, "payload": {"file_path": "/project/utils.py", "entity_name": "helper"}},
            {"id": "3", "score": 0.80, "payload": {"file_path": "a.py", "entity_name": "in_a", "summary": "sum"}}]

    def smocks():
        q = AsyncMock()
        q.search = AsyncMock(return_value=hits)
        e = AsyncMock()
        e.embed = AsyncMock(return_value=[0.5, 0.5])
        return q, e
    a11 = {}
    q, e = smocks()
    r = await R.indexer.VectorSearcher(q, e).search_code("hello world", limit=7)
    a11["plain"] = {"store": call_log(q), "embedder": call_log(e), "results": [dataclasses.asdict(x) for x in r]}
    q, e = smocks()
    await R.indexer.VectorSearcher(q, e).search_code("hello", language="python", entity_type="function", project_name="my-project")
    a11["filters"] = {"store": call_log(q)}
    q, e = smocks()
    r = await R.indexer.VectorSearcher(q, e).search_summaries("greeting", entity_type="class")
    a11["summaries"] = {"store": call_log(q), "results": [dataclasses.asdict(x) for x in r]}
    q, e = smocks()
    e.embed.side_effect = Exception("API Error")
    try:
        await R.indexer.VectorSearcher(q, e).search_code("test query")
    except R.errors.IndexingError as ex:
        a11["error"] = {"stage": ex.stage, "str": str(ex)}
    out["a11_indexer_searcher"] = a11

    a12 = {}
    q, e = smocks()
    r = await R.vsearch.VectorSearcher(q, e).search_code("find auth", limit=4)
    a12["plain"] = {"store": call_log(q), "embedder": call_log(e), "results": r}
    q, e = smocks()
    await R.vsearch.VectorSearcher(q, e).search_code("find auth", language="python")
    a12["language"] = {"store": call_log(q)}
    q, e = smocks()
    r = await R.vsearch.VectorSearcher(q, e).search_summaries("what", limit=3, project_name="proj")
    a12["summaries"] = {"store": call_log(q), "results": r}
    q, e = smocks()
    r = await R.vsearch.VectorSearcher(q, e).find_similar_code("def f(): pass", limit=1, exclude_file="a.py")
    a12["similar_exclude"] = {"store": call_log(q), "results": r}
    q, e = smocks()
    r = await R.vsearch.VectorSearcher(q, e).find_similar_code("def f(): pass", limit=2)
    a12["similar_plain"] = {"store": call_log(q), "results": r}
    errs = {}
    for label, coro in (("blank_code", lambda s: s.search_code("   ")), ("blank_summary", lambda s: s.search_summaries("")),
                        ("blank_similar", lambda s: s.find_similar_code("\n"))):
        q, e = smocks()
        try:
            await coro(R.vsearch.VectorSearcher(q, e))
        except R.errors.QueryError as ex:
            errs[label] = str(ex)
    for label, exc, call in (("embed_code", R.errors.EmbeddingError("boom"), lambda s: s.search_code("x")),
                             ("store_code", R.errors.VectorStoreError("down"), lambda s: s.search_code("x")),
                             ("store_summary", R.errors.VectorStoreError("down"), lambda s: s.search_summaries("x")),
                             ("embed_similar", R.errors.EmbeddingError("boom"), lambda s: s.find_similar_code("x")),
                             ("store_similar", R.errors.VectorStoreError("down"), lambda s: s.find_similar_code("x"))):
        q, e = smocks()
        if label.startswith("embed"):
            e.embed.side_effect = exc
        else:
            q.search.side_effect = exc
        try:
            await call(R.vsearch.VectorSearcher(q, e))
        except R.errors.QueryError as ex:
            errs[label] = str(ex)
    a12["errors"] = errs
    out["a12_query_searcher"] = a12
    return out

# ---------------------------------------------------------------------------------------------- chunking
def chunker_scenarios():
    """Parsed files (as plain dicts) x chunker parameters.  The token counter is the stand-in installed above
    (whitespace split): tiktoken's cl100k table is not in the image, so the goldens pin the chunking ALGORITHM
    (entity formatting, line packing, overlap carry, naming, line numbers), with the counter as an input."""
    import random
    rng = random.Random(20260104)
    words = ["alpha", "beta", "gamma", "delta", "x", "=", "return", "self.value", "(", ")", "if", "else:", "for", "in", "range(10):"]

    def body(n_lines, lo=1, hi=9, blank_every=0):
        lines = []
        for i in range(n_lines):
            if blank_every and i % blank_every == blank_every - 1:
                lines.append("")
            else:
                lines.append("    " + " ".join(rng.choice(words) for _ in range(rng.randint(lo, hi))))
        return "\n".join(lines)

    def ent(name, code, etype="function", sig=None, doc=None, start=1, children=(), qn=None):
        return dict(type=etype, name=name, qualified_name=qn or f"pkg.mod.{name}", signature=sig, docstring=doc, code=code,
                    start_line=start, end_line=start + code.count("\n"), children=list(children))

    def pfile(path, content, entities, language="python"):
        return dict(path=path, language=language, content_hash=f"h-{len(content)}", content=content, entities=entities)

    big = body(120, blank_every=7)
    huge_line = "    " + " ".join("tok%d" % i for i in range(90))
    files = {
        "small_entities": pfile("src/app/small.py", "irrelevant", [
            ent("load", body(4), sig="def load(path):", doc="Load a file.", start=3),
            ent("Repo", body(6), etype="class", sig="class Repo:", start=20,
                children=[ent("get", body(3), etype="method", sig="def get(self, k):", start=22, qn="pkg.mod.Repo.get"),
                          ent("put", body(5), etype="method", sig=None, doc="Store.", start=27, qn="pkg.mod.Repo.put")]),
        ]),
        "one_large_entity": pfile("src/app/large.py", "irrelevant", [ent("pipeline", big, sig="def pipeline(cfg):", doc="Run it.", start=40)]),
        "mixed": pfile("src/app/mixed.ts", "irrelevant", [
            ent("tiny", "    return 1", sig="function tiny()", start=1),
            ent("wide", body(60, lo=6, hi=12), sig="function wide(a, b)", start=10),
            ent("empty_code", "", sig="function empty_code()", start=90),
            ent("doc_only", body(2), sig=None, doc="Only a docstring\nover two lines.", start=95),
        ], language="typescript"),
        "line_longer_than_max": pfile("src/app/longline.py", "irrelevant", [
            ent("dense", "\n".join([body(3), huge_line, body(2), huge_line, huge_line, body(4)]), sig="def dense():", start=7)]),
        "no_entities_fallback": pfile("docs/notes.py", body(75, blank_every=5), []),
        "no_entities_short": pfile("docs/short.py", "x = 1\ny = 2\n", []),
        "blank_content": pfile("docs/blank.py", "  \n\n\t\n", []),
        "trailing_newlines": pfile("src/app/trail.py", "irrelevant", [ent("padded", body(30) + "\n\n\n", sig="def padded():", start=5)]),
    }
    params = [dict(max_tokens=None, overlap_tokens=None), dict(max_tokens=60, overlap_tokens=15), dict(max_tokens=60, overlap_tokens=0),
              dict(max_tokens=25, overlap_tokens=40), dict(max_tokens=40, overlap_tokens=5), dict(max_tokens=12, overlap_tokens=3)]
    return files, params


def compact_chunk(c):
    """Long contents are frozen as sha1 + length (an equally strong pin at a fraction of the fixture size)."""
    import hashlib
    d = dataclasses.asdict(c)
    if len(d["content"]) > 160:
        text = d.pop("content")
        d["content_sha1"], d["content_len"] = hashlib.sha1(text.encode()).hexdigest(), len(text)
    return d


def run_chunker(R):
    files, params = chunker_scenarios()
    Lang, EType = R.types.Language, R.types.EntityType

    def build_entity(d):
        return R.pmodels.CodeEntity(type=EType(d["type"]), name=d["name"], qualified_name=d["qualified_name"], signature=d["signature"],
                                    docstring=d["docstring"], code=d["code"], start_line=d["start_line"], end_line=d["end_line"],
                                    children=[build_entity(c) for c in d["children"]])

    cases = []
    for fname, f in files.items():
        info = R.pmodels.FileInfo(path=Path(f["path"]), relative_path=f["path"], language=Lang(f["language"]), content_hash=f["content_hash"],
                                  size_bytes=len(f["content"]), line_count=f["content"].count("\n") + 1)
        parsed = R.pmodels.ParsedFile(file_info=info, content=f["content"], entities=[build_entity(e) for e in f["entities"]])
        for p in params:
            for project in (None, "demo"):
                chunks = R.chunker.CodeChunker(**p).chunk_file(parsed, project_name=project)
                cases.append({"file": fname, "params": p, "project_name": project, "chunks": [compact_chunk(c) for c in chunks]})
                if p["max_tokens"] != 60:           # the project_name pass-through needs only one parameter set
                    break
    return {"generator": "tests/golden/gen_goldens.py", "reference": "src/lattice/embeddings/chunker.py:40-217",
            "token_counter": "len(text.split()) stand-in for tiktoken cl100k_base (absent offline)",
            "settings": {"chunk_max_tokens": 1000, "chunk_overlap_tokens": 200}, "files": files, "cases": cases}


def main():
    R = load_reference()
    OUT.mkdir(parents=True, exist_ok=True)
    scen = ranking_scenarios(R)
    ranking = {"generator": "tests/golden/gen_goldens.py", "reference": "src/lattice/query/ranking/{models,scorer,ranker,utils}.py",
               "scenarios": [dict(s, expected=run_ranking(R, s)) for s in scen]}
    (OUT / "ranking_reference.json").write_text(json.dumps(ranking, indent=1, sort_keys=True, default=str))
    # seeded random scenarios (inputs regenerated by tests/ranking_cases.py; only digests + a readable head are stored)
    import hashlib
    sys.path.insert(0, str(OUT.parent))
    import ranking_cases
    rnd = []
    for sc in ranking_cases.scenarios():
        exp = json.loads(json.dumps(run_ranking(R, sc), default=str))
        rnd.append({"name": sc["name"], "n": len(exp["ranked"]), "digest": hashlib.sha1(json.dumps(exp, sort_keys=True).encode()).hexdigest(),
                    "head": [[r["entity_name"], r["file_path"], r["start_line"], r["source"], r["final_score"]] for r in exp["ranked"][:4]]})
    (OUT / "ranking_random_reference.json").write_text(json.dumps({"generator": "tests/golden/gen_goldens.py + tests/ranking_cases.py", "seed": ranking_cases.SEED,
                                                                   "cases": rnd}, indent=0, sort_keys=True))
    plans = {}
    planner = R.planner.QueryPlanner.__new__(R.planner.QueryPlanner)
    for qtext in ("how does verify_password work in UserRepository", "what calls process_payment", "where is the ConfigLoader",
                  "find code similar to `retry_with_backoff`", "explain the architecture"):
        pl = planner._fallback_plan(qtext)
        plans[qtext] = {"intent": pl.primary_intent.value, "entities": [e.name for e in pl.entities]}
    shapes = asyncio.run(run_callshapes(R))
    shapes["fallback_plans"] = plans
    (OUT / "callshapes_reference.json").write_text(json.dumps(shapes, indent=1, sort_keys=True, default=str))
    (OUT / "chunker_reference.json").write_text(json.dumps(run_chunker(R), sort_keys=True, separators=(",", ":"), default=str))
    print("wrote", OUT / "ranking_reference.json", "and", OUT / "callshapes_reference.json")


if __name__ == "__main__":
    main()
