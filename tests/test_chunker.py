"""CodeChunker against the reference's own chunker (src/lattice/embeddings/chunker.py:40-217).

tests/golden/chunker_reference.json holds parsed files, chunker parameters and the chunks the reference's
CodeChunker.chunk_file produced for them (tests/golden/gen_goldens.py runs the reference class by path).  tiktoken's
cl100k_base table is not available offline, so the generator installs ``len(text.split())`` as the token counter
and this test hands the same counter to our chunker: what is pinned is the algorithm -- entity formatting, line
packing, overlap carry, chunk naming, line numbers, the whole-file fallback -- with the counter as an input."""
import hashlib
import json
import os
import types
from pathlib import Path

import pytest

import coderag_amd  # noqa: F401
from coderag_amd import indexer as indexer_mod

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chunker_reference.json")))


class ParsedFileStub:
    """What chunk_file reads from a ParsedFile (parsing/models.py:41-60), including its depth-first
    ``all_entities`` walk that pops from the end of a stack."""

    def __init__(self, spec):
        self.file_info = types.SimpleNamespace(path=Path(spec["path"]), language=types.SimpleNamespace(value=spec["language"]),
                                               content_hash=spec["content_hash"])
        self.content = spec["content"]
        self.entities = [self._entity(e) for e in spec["entities"]]

    @classmethod
    def _entity(cls, d):
        return types.SimpleNamespace(type=types.SimpleNamespace(value=d["type"]), name=d["name"], qualified_name=d["qualified_name"],
                                     signature=d["signature"], docstring=d["docstring"], code=d["code"], start_line=d["start_line"],
                                     end_line=d["end_line"], children=[cls._entity(c) for c in d["children"]])

    @property
    def all_entities(self):
        out, stack = [], list(self.entities)
        while stack:
            entity = stack.pop()
            out.append(entity)
            stack.extend(entity.children)
        return out


def as_golden(chunk):
    d = {"content": chunk.content, "file_path": chunk.file_path, "entity_type": chunk.entity_type, "entity_name": chunk.entity_name,
         "language": chunk.language, "start_line": chunk.start_line, "end_line": chunk.end_line, "graph_node_id": chunk.graph_node_id,
         "content_hash": chunk.content_hash, "project_name": chunk.project_name}
    if len(d["content"]) > 160:
        text = d.pop("content")
        d["content_sha1"], d["content_len"] = hashlib.sha1(text.encode()).hexdigest(), len(text)
    return d


def test_goldens_cover_the_branches():
    counts = {}
    for case in GOLD["cases"]:
        counts.setdefault(case["file"], set()).add(len(case["chunks"]))
    assert counts["blank_content"] == {0} and counts["no_entities_short"] == {1}
    assert max(counts["one_large_entity"]) > 50 and 1 in counts["one_large_entity"]        # both the whole-entity and the split branch
    assert any("_part" in c["entity_name"] for case in GOLD["cases"] for c in case["chunks"])
    assert any(case["params"]["overlap_tokens"] == 0 for case in GOLD["cases"])              # the falsy-overlap default


@pytest.mark.parametrize("case", GOLD["cases"], ids=lambda c: f"{c['file']}-{c['params']['max_tokens']}-{c['params']['overlap_tokens']}"
                                                             f"-{c['project_name']}")
def test_chunks_match_reference(case):
    chunker = indexer_mod.CodeChunker(**case["params"], encode=str.split)
    got = chunker.chunk_file(ParsedFileStub(GOLD["files"][case["file"]]), project_name=case["project_name"])
    assert [as_golden(c) for c in got] == case["chunks"]


def test_defaults_follow_settings():
    c = indexer_mod.CodeChunker(encode=str.split)
    assert (c.max_tokens, c.overlap_tokens) == (GOLD["settings"]["chunk_max_tokens"], GOLD["settings"]["chunk_overlap_tokens"])
    assert indexer_mod.CodeChunker(max_tokens=60, overlap_tokens=0, encode=str.split).overlap_tokens == 200      # chunker.py:49
    assert c.count_tokens("a b  c\n") == 3
