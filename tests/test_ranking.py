"""HybridRanker restatement against outputs captured from the reference's own ranking code
(tests/golden/ranking_reference.json, produced by tests/golden/gen_goldens.py)."""
import dataclasses
import json
import os

import pytest

import coderag_amd  # noqa: F401
from coderag_amd.query_types import ExtractedEntity, GraphContext, GraphNode, QueryIntent, QueryPlan
from coderag_amd.ranking import HybridRanker, RankingConfig, ranked_results_to_search_results

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ranking_reference.json")))
ROLES = ("primary_entities", "callers", "callees", "parent_classes", "child_classes", "methods")


def build_inputs(s):
    plan = QueryPlan(original_query=s["name"], primary_intent=QueryIntent(s["intent"]),
                     entities=[ExtractedEntity(name=e) for e in s["entities"]])
    ctx = GraphContext(**{role: [GraphNode(**n) for n in s["graph"].get(role, [])] for role in ROLES})
    return plan, ctx, [dict(v) for v in s["vector"]], s["centrality"]


@pytest.mark.parametrize("scenario", GOLD["scenarios"], ids=lambda s: s["name"])
def test_ranker_matches_reference_output(scenario):
    plan, ctx, vec, cent = build_inputs(scenario)
    ranked = HybridRanker().rank_results(plan, ctx, vec, cent)
    got = json.loads(json.dumps([dataclasses.asdict(r) for r in ranked], default=str))
    assert got == scenario["expected"]["ranked"]            # every score, order, signal, source -- floats exact
    flat = json.loads(json.dumps(ranked_results_to_search_results(ranked), default=str))
    assert flat == scenario["expected"]["flattened"]


def test_documented_example_values():
    s = next(x for x in GOLD["scenarios"] if x["name"] == "survey_example")
    ranked = HybridRanker().rank_results(*build_inputs(s))
    assert [r.final_score for r in ranked] == [0.8700000000000001, 0.43000000000000005]
    assert all(r.source == "vector" for r in ranked)
    assert set(ranked[0].signal_scores) == {"vector_similarity", "query_entity_match", "centrality", "code_quality"}


def test_caps_and_config_override():
    s = next(x for x in GOLD["scenarios"] if x["name"] == "total_cap")
    plan, ctx, vec, cent = build_inputs(s)
    assert len(HybridRanker().rank_results(plan, ctx, vec, cent)) == 50
    small = HybridRanker(RankingConfig(max_per_file=1, max_total=7)).rank_results(plan, ctx, vec, cent)
    assert len(small) == 7 and len({r.file_path for r in small}) == 7
    w = RankingConfig().weights_for(QueryIntent.FIND_IMPLEMENTATIONS)   # intent without a row keeps defaults
    assert (w["graph_weight"], w["vector_weight"]) == (0.5, 0.5)
    assert RankingConfig().weights_for("find_call_chain")["graph_weight"] == 0.9   # plain strings work too


# ---- 160 seeded random scenarios (tests/ranking_cases.py makes the inputs; the reference's ranker made the expected digests)
import hashlib

from tests import ranking_cases

RANDOM_GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ranking_random_reference.json")))


def test_random_scenarios_cover_the_branches():
    cases = RANDOM_GOLD["cases"]
    assert len(cases) == ranking_cases.N_SCENARIOS and RANDOM_GOLD["seed"] == ranking_cases.SEED
    sources = {row[3] for c in cases for row in c["head"]}
    assert sources == {"graph", "vector", "hybrid"}
    assert min(c["n"] for c in cases) <= 2 and max(c["n"] for c in cases) == 35      # 7 files x the per-file cap of 5


@pytest.mark.parametrize("case", RANDOM_GOLD["cases"], ids=lambda c: c["name"])
def test_ranker_matches_reference_on_random_scenarios(case):
    sc = ranking_cases.scenario(int(case["name"].split("_")[1]))
    plan, ctx, vec, cent = build_inputs(sc)
    ranked = HybridRanker().rank_results(plan, ctx, vec, cent)
    got = json.loads(json.dumps({"ranked": [dataclasses.asdict(r) for r in ranked], "flattened": ranked_results_to_search_results(ranked)}, default=str))
    head = [[r["entity_name"], r["file_path"], r["start_line"], r["source"], r["final_score"]] for r in got["ranked"][:4]]
    assert (len(got["ranked"]), head) == (case["n"], case["head"])
    assert hashlib.sha1(json.dumps(got, sort_keys=True).encode()).hexdigest() == case["digest"]     # every field of every result


# ---- 120 seeded vector-only scenarios (what the device re-rank decides); expected rows from the reference's ranker
VO_GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ranking_vector_only_reference.json")))
VO_SIGNALS = ("vector_similarity", "query_entity_match", "centrality", "code_quality")


def vector_only_rows(ranked):
    return [[r.entity_name, r.file_path, r.start_line, getattr(r.source, "value", r.source), r.final_score] + [r.signal_scores.get(s) for s in VO_SIGNALS]
            for r in ranked]


@pytest.mark.parametrize("case", VO_GOLD["cases"], ids=lambda c: c["name"])
def test_host_ranker_matches_reference_on_vector_only_scenarios(case):
    sc = ranking_cases.vector_only_scenario(int(case["name"].split("_")[2]))
    ranked = HybridRanker().rank_results(*build_inputs(sc))
    assert vector_only_rows(ranked) == case["rows"]


def test_vector_only_scenarios_cover_the_branches():
    cases = VO_GOLD["cases"]
    assert len(cases) == ranking_cases.N_VECTOR_ONLY and VO_GOLD["seed"] == ranking_cases.VSEED
    assert {row[3] for c in cases for row in c["rows"]} == {"vector", "hybrid"}
    assert max(c["n"] for c in cases) == 50 and min(c["n"] for c in cases) == 1             # total cap reached; single hit
    assert any(row[6] == 0.5 for c in cases for row in c["rows"]) and any(row[6] == 1.0 for c in cases for row in c["rows"])
    assert any(0.0 < row[7] < 1.0 for c in cases for row in c["rows"]) and any(row[7] == 1.0 for c in cases for row in c["rows"])
