"""Indexing and the dataclass-flavoured searcher of ``src/lattice/embeddings/indexer.py`` + the chunk payload
schema of ``src/lattice/embeddings/chunker.py:12-37``.

``VectorIndexer`` keeps the reference's control flow exactly (hash-skip unless forced -> delete by file ->
chunk -> embed -> uuid4 ids -> one upsert; every failure wrapped in ``IndexingError``): it is duck-typed over
the store and the embedder, so it runs unchanged on ``HipVectorStore`` + ``HipUniXcoderProvider``.
"""

from __future__ import annotations

import asyncio
import logging
import re
import uuid
from collections.abc import Callable
from dataclasses import asdict, dataclass
from typing import Any

from .errors import IndexingError
from .store import CollectionName

logger = logging.getLogger(__name__)

CHUNK_NAME_SEPARATOR = "_part"


@dataclass
class CodeChunk:
    content: str
    file_path: str
    entity_type: str
    entity_name: str
    language: str
    start_line: int
    end_line: int
    graph_node_id: str | None = None
    content_hash: str | None = None
    project_name: str | None = None

    def to_payload(self) -> dict:
        """Payload stored next to the vector (chunker.py:24-37): every field, ``content`` included."""
        return {"file_path": self.file_path, "entity_type": self.entity_type, "entity_name": self.entity_name,
                "language": self.language, "start_line": self.start_line, "end_line": self.end_line, "content": self.content,
                "graph_node_id": self.graph_node_id, "content_hash": self.content_hash, "project_name": self.project_name}


_WORDISH = re.compile(r"\w+|[^\w\s]")


class CodeChunker:
    """Entity-wise chunking with line-granular splitting and overlap (chunker.py:40-217).

    The reference counts cl100k_base tokens with ``tiktoken``; when that package is importable it is used, so
    chunk boundaries are identical.  Without it (this image) a word/punctuation count stands in -- a NEXT row
    of SURVEY.md section 8f, not part of the measured hot path."""

    def __init__(self, max_tokens: int | None = None, overlap_tokens: int | None = None, encoding_name: str = "cl100k_base",
                 *, encode: Callable[[str], list] | None = None):
        # both fall back on a falsy value, so overlap_tokens=0 means "the configured default" (chunker.py:47-49)
        self.max_tokens = max_tokens or 1000
        self.overlap_tokens = overlap_tokens or 200
        self._count = None
        if encode is not None:              # an explicit token counter (tests pin the algorithm with the goldens' one)
            self._encode = encode
            return
        try:
            import tiktoken
            self._encode = tiktoken.get_encoding(encoding_name).encode
        except Exception:
            self._encode = _WORDISH.findall
            try:        # the same count for ASCII text from libcoderag_tok.so (the regex costs ~70 us per chunk: the largest host cost of indexing)
                from .tokenizer_native import count_wordish_ascii
                count_wordish_ascii("probe text")
                self._count = count_wordish_ascii
            except Exception:   # noqa: BLE001 -- library not built: the regex alone
                self._count = None

    def count_tokens(self, text: str) -> int:
        if self._count is not None:
            n = self._count(text)
            if n >= 0:
                return n
        return len(self._encode(text))

    def chunk_file(self, parsed_file, project_name: str | None = None) -> list[CodeChunk]:
        info = parsed_file.file_info
        file_path, language = str(info.path), getattr(info.language, "value", info.language)
        common = dict(file_path=file_path, language=language, content_hash=info.content_hash, project_name=project_name)
        chunks: list[CodeChunk] = []
        for entity in parsed_file.all_entities:
            head = [part for part in (entity.signature, f'"""{entity.docstring}"""' if entity.docstring else None) if part]
            text = "\n".join(head + [entity.code])          # the code is always appended, even when empty (chunker.py:127-134)
            etype = getattr(entity.type, "value", entity.type)
            if self.count_tokens(text) <= self.max_tokens:
                chunks.append(CodeChunk(content=text, entity_type=etype, entity_name=entity.qualified_name,
                                        start_line=entity.start_line, end_line=entity.end_line,
                                        graph_node_id=entity.qualified_name, **common))
            else:
                chunks.extend(self._split(text, etype, entity.qualified_name, entity.start_line, common))
        if not chunks and parsed_file.content.strip():
            chunks.extend(self._split(parsed_file.content, "file", info.path.name, 1, common))
        return chunks

    def _split(self, text: str, entity_type: str, entity_name: str, first_line: int, common: dict) -> list[CodeChunk]:
        out: list[CodeChunk] = []
        held: list[str] = []
        held_tokens = 0
        start = first_line

        def emit(final: bool) -> None:
            name = entity_name if (final and not out) else f"{entity_name}{CHUNK_NAME_SEPARATOR}{len(out) + 1}"
            out.append(CodeChunk(content="\n".join(held), entity_type=entity_type, entity_name=name, start_line=start,
                                 end_line=start + len(held) - 1, graph_node_id=entity_name, **common))

        for i, line in enumerate(text.split("\n")):
            cost = self.count_tokens(line + "\n")
            if held and held_tokens + cost > self.max_tokens:
                emit(final=False)
                tail: list[str] = []
                tail_tokens = 0
                for prev in reversed(held):          # carry trailing lines worth <= overlap_tokens into the next chunk
                    c = self.count_tokens(prev + "\n")
                    if tail_tokens + c > self.overlap_tokens:
                        break
                    tail.insert(0, prev)
                    tail_tokens += c
                held, held_tokens = tail, tail_tokens
                start = first_line + i - len(tail)
            held.append(line)
            held_tokens += cost
        if held:
            emit(final=True)
        return out


@dataclass
class CodeSearchResult:
    score: float
    file_path: str
    entity_type: str
    entity_name: str
    content: str
    start_line: int
    end_line: int


@dataclass
class SummarySearchResult:
    score: float
    file_path: str
    entity_type: str
    entity_name: str
    summary: str


class VectorIndexer:
    GROUP, GROUP_FIRST = 4096, 512       # chunks per pipelined group of index_files_batched: the largest / the first

    def __init__(self, qdrant, embedder, chunker=None):
        self.qdrant = qdrant
        self.embedder = embedder
        self.chunker = chunker or CodeChunker()

    async def index_file(self, parsed_file, progress_callback: Callable[[int, int], None] | None = None,
                         force: bool = False, project_name: str | None = None) -> int:
        """indexer.py:46-94."""
        try:
            file_path = str(parsed_file.file_info.path)
            if not force and not await self._needs_indexing(file_path, parsed_file.file_info.content_hash):
                logger.debug(f"Skipping unchanged file: {file_path}")
                return 0
            await self.qdrant.delete(CollectionName.CODE_CHUNKS.value, {"file_path": file_path})
            chunks = self.chunker.chunk_file(parsed_file, project_name=project_name)
            if not chunks:
                logger.debug(f"No chunks generated for file: {file_path}")
                return 0
            in_store = self._embed_in_store()
            if in_store is not None:
                # one process per shard: the store routes the rows, every rank embeds its own share, and an encoder failure on ONE
                # rank is agreed by all inside the upsert (a rank that raised here, before the store's collectives, would leave
                # the others waiting in them)
                texts = [c.content for c in chunks]
                await self.qdrant.upsert(CollectionName.CODE_CHUNKS.value, [str(uuid.uuid4()) for _ in chunks], None,
                                         [c.to_payload() for c in chunks], texts=texts, embed=in_store)
                if progress_callback:
                    progress_callback(len(texts), len(texts))
            else:
                vectors = await self.embedder.embed_with_progress([c.content for c in chunks], progress_callback=progress_callback)
                await self.qdrant.upsert(collection=CollectionName.CODE_CHUNKS.value, ids=[str(uuid.uuid4()) for _ in chunks],
                                         vectors=vectors, payloads=[c.to_payload() for c in chunks])
            logger.info(f"Indexed {len(chunks)} chunks from {file_path}")
            return len(chunks)
        except Exception as e:
            raise IndexingError(f"Failed to index file {parsed_file.file_info.path}", stage="file_indexing", cause=e)

    def _embed_in_store(self):
        """The provider's synchronous embed callable when the store shards across PROCESSES (backend 'dist'), else None."""
        backend = getattr(self.qdrant, "__dict__", {}).get("_shard_backend")       # (an instance attribute of HipVectorStore; other stores have none)
        if not (isinstance(backend, str) and backend == "dist"):
            return None
        return getattr(getattr(self.embedder, "provider", None), "embed_texts_sync", None)

    async def index_files(self, parsed_files: list, progress_callback: Callable[[int, int], None] | None = None,
                          project_name: str | None = None) -> int:
        """Sequential; a failing file is logged and skipped (indexer.py:96-119)."""
        total = 0
        for done, parsed_file in enumerate(parsed_files, start=1):
            try:
                total += await self.index_file(parsed_file, project_name=project_name)
            except IndexingError as e:
                logger.error(f"Failed to index file: {e}")
                continue
            if progress_callback:
                progress_callback(done, len(parsed_files))
        logger.info(f"Indexed total of {total} chunks from {len(parsed_files)} files")
        return total

    async def index_files_batched(self, parsed_files: list, progress_callback: Callable[[int, int], None] | None = None,
                                  project_name: str | None = None, force: bool = False, embed_in_store: bool | None = None) -> int:
        """The same outcome as :meth:`index_files` -- per file: skip when unchanged, delete its old chunks, chunk, embed, upsert
        under fresh uuid4 ids with ``to_payload()`` payloads -- as ONE pass over all files instead of one round trip chain per
        file (the reference's flow, indexer.py:96-119 over :meth:`index_file`, spends its time in ~5 awaited hops per file:
        3.2 k chunks/s against an encoder that embeds 25 k): one update check, one delete job, all files chunked, ONE coalesced
        embedding submission handed over as a float32 array, ONE upsert.  ``embed_in_store`` (default: when the store shards
        across processes): the store routes the rows first and every process embeds only its own shards' texts
        (``HipVectorStore.upsert(vectors=None, texts=..., embed=...)``).  If anything in the batch fails, the files go through
        the sequential flow, so a bad file still loses only itself."""
        files = list(parsed_files)
        if not files:
            return 0
        in_flight = None
        try:
            name = CollectionName.CODE_CHUNKS.value
            paths = [str(f.file_info.path) for f in files]
            if force:
                todo = files
            else:
                many = getattr(self.qdrant, "files_need_update", None)
                pairs = [(p, f.file_info.content_hash) for p, f in zip(paths, files)]
                needs = await many(name, pairs) if many is not None else [await self._needs_indexing(p, h) for p, h in pairs]
                todo = [f for f, need in zip(files, needs) if need]
            if todo:
                todo_paths = [str(f.file_info.path) for f in todo]
                many_del = getattr(self.qdrant, "delete_files", None)
                if many_del is not None:
                    await many_del(name, todo_paths)
                else:
                    for p in todo_paths:
                        await self.qdrant.delete(name, {"file_path": p})
            sync_embed = getattr(getattr(self.embedder, "provider", None), "embed_texts_sync", None)
            if embed_in_store is None:
                embed_in_store = self._embed_in_store() is not None
            to_array = getattr(self.embedder, "embed_array", None)

            async def embed_and_store(chunks):
                texts = [c.content for c in chunks]
                ids = [str(uuid.uuid4()) for _ in chunks]
                payloads = [c.to_payload() for c in chunks]
                if embed_in_store and sync_embed is not None:
                    await self.qdrant.upsert(name, ids, None, payloads, texts=texts, embed=sync_embed)
                    return
                vectors = await to_array(texts) if to_array is not None else await self.embedder.embed_batch(texts, batch_size=len(texts))
                await self.qdrant.upsert(collection=name, ids=ids, vectors=vectors, payloads=payloads)

            # Chunking is host work (the token counter), embedding is device work driven from the provider's thread, the upsert
            # runs on the store's: groups of ~GROUP chunks go down the three stages one behind the other, so the encoder is fed
            # while the next group is being cut (14 k chunks: chunking 0.2 s + tables 0.1 s beside 1.0 s of encoder).
            # (the chunking loop holds the interpreter lock in 5 ms slices by default; the provider's thread needs it for a few
            # microseconds between kernel launches and would wait out every slice: hand it over a hundred times more often)
            import sys
            switch = sys.getswitchinterval()
            sys.setswitchinterval(min(switch, 5e-5))
            total, group, want = 0, [], self.GROUP_FIRST
            for f in todo:
                group.extend(self.chunker.chunk_file(f, project_name=project_name))
                if len(group) >= want:
                    if in_flight is not None:
                        await in_flight
                    in_flight, total, group = asyncio.ensure_future(embed_and_store(group)), total + len(group), []
                    want = min(self.GROUP, want * 4)            # a small first group (the encoder starts early), then large ones (fewer, fuller submissions)
                    await asyncio.sleep(0)                      # (let the new task reach its first await: the executor hand-over)
            sys.setswitchinterval(switch)
            if in_flight is not None:
                await in_flight
            if group:
                await embed_and_store(group)
                total += len(group)
            if progress_callback:
                for done in range(1, len(files) + 1):
                    progress_callback(done, len(files))
            logger.info(f"Indexed total of {total} chunks from {len(files)} files ({len(todo)} changed)")
            return total
        except Exception as e:  # noqa: BLE001
            logger.error(f"Batched indexing failed ({e!r}); indexing the files one by one")
            import sys
            if sys.getswitchinterval() < 1e-3:
                sys.setswitchinterval(0.005)
            if in_flight is not None and not in_flight.done():      # (a group still on its way must land, or fail, before its files are looked at again)
                try:
                    await in_flight
                except Exception:  # noqa: BLE001
                    pass
            total = 0
            for done, parsed_file in enumerate(files, start=1):          # index_files' loop (indexer.py:104-116), `force` carried along
                try:
                    total += await self.index_file(parsed_file, force=force, project_name=project_name)
                except IndexingError as err:
                    logger.error(f"Failed to index file: {err}")
                    continue
                if progress_callback:
                    progress_callback(done, len(files))
            return total

    async def index_summary(self, file_path: str, entity_type: str, entity_name: str, summary: str,
                            graph_node_id: str | None = None) -> None:
        """One embed + one upsert into ``summaries`` (indexer.py:121-152)."""
        try:
            vector = await self.embedder.embed(summary)
            await self.qdrant.upsert(collection=CollectionName.SUMMARIES.value, ids=[str(uuid.uuid4())], vectors=[vector],
                                     payloads=[{"file_path": file_path, "entity_type": entity_type, "entity_name": entity_name,
                                                "summary": summary, "graph_node_id": graph_node_id}])
            logger.info(f"Indexed summary for {entity_name} in {file_path}")
        except Exception as e:
            raise IndexingError(f"Failed to index summary for {entity_name}", stage="summary_indexing", cause=e)

    async def _needs_indexing(self, file_path: str, content_hash: str) -> bool:
        return await self.qdrant.file_needs_update(CollectionName.CODE_CHUNKS.value, file_path, content_hash)


def _only_set(**kw: Any) -> dict[str, Any] | None:
    chosen = {k: v for k, v in kw.items() if v}
    return chosen or None


class VectorSearcher:
    """Dataclass-returning searcher (indexer.py:162-257); every failure -> ``IndexingError`` (quirk Q5)."""

    def __init__(self, qdrant, embedder):
        self.qdrant = qdrant
        self.embedder = embedder

    async def _run(self, collection: str, query: str, limit: int, filters: dict | None, stage: str, what: str):
        try:
            vector = await self.embedder.embed(query)
            return await self.qdrant.search(collection=collection, query_vector=vector, limit=limit, filters=filters)
        except Exception as e:
            logger.error(f"{what} search failed: {e}")
            raise IndexingError(f"Failed to search {what.lower()} for query: {query}", stage=stage, cause=e)

    async def search_code(self, query: str, limit: int = 10, language: str | None = None, entity_type: str | None = None,
                          project_name: str | None = None) -> list[CodeSearchResult]:
        hits = await self._run(CollectionName.CODE_CHUNKS.value, query, limit,
                               _only_set(language=language, entity_type=entity_type, project_name=project_name),
                               "code_search", "Code")
        return self._format_code_results(hits)

    async def search_summaries(self, query: str, limit: int = 10, entity_type: str | None = None) -> list[SummarySearchResult]:
        hits = await self._run(CollectionName.SUMMARIES.value, query, limit, _only_set(entity_type=entity_type),
                               "summary_search", "Summaries")
        return self._format_summary_results(hits)

    @staticmethod
    def _format_code_results(results: list[dict]) -> list[CodeSearchResult]:
        out = []
        for hit in results:
            p = hit["payload"]
            out.append(CodeSearchResult(score=hit["score"], file_path=p.get("file_path", ""), entity_type=p.get("entity_type", ""),
                                        entity_name=p.get("entity_name", ""), content=p.get("content", ""),
                                        start_line=p.get("start_line", 0), end_line=p.get("end_line", 0)))
        return out

    @staticmethod
    def _format_summary_results(results: list[dict]) -> list[SummarySearchResult]:
        out = []
        for hit in results:
            p = hit["payload"]
            out.append(SummarySearchResult(score=hit["score"], file_path=p.get("file_path", ""),
                                           entity_type=p.get("entity_type", ""), entity_name=p.get("entity_name", ""),
                                           summary=p.get("summary", "")))
        return out
