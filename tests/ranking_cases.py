"""Seeded random scenarios for the hybrid ranker -- plain dictionaries in the shape tests/golden/gen_goldens.py feeds the
REFERENCE's HybridRanker and tests/test_ranking.py feeds ours.  Only inputs are made here; the expected outputs in
tests/golden/ranking_random_reference.json come from the reference's code.  Collisions are deliberate: a small pool of
names / files / lines makes graph nodes and vector hits meet on the merge key, repeat files past the per-file cap, tie scores."""
import random

INTENTS = ["find_callers", "find_callees", "find_call_chain", "find_hierarchy", "find_implementations", "find_usages", "find_dependencies",
           "find_dependents", "locate_entity", "locate_file", "explain_implementation", "explain_relationship", "explain_data_flow",
           "explain_architecture", "find_similar", "search_functionality", "search_pattern"]
NAMES = ["UserRepository", "verify_password", "login", "refresh", "hash_password", "BaseRepository", "CachedUserRepository", "parse",
         "parse_config", "Config", "config", "load", "Loader", "retry", "retry_with_backoff", "main", "", "Ünïcode", "x"]
FILES = ["src/a.py", "src/auth.py", "src/repo.py", "src/crypto.py", "src/base.py", "lib/x.ts", "lib/y.ts"]
ROLES = ("primary_entities", "callers", "callees", "parent_classes", "child_classes", "methods")
N_SCENARIOS, SEED = 160, 20260104


def _node(rng, depth_ok):
    name = rng.choice(NAMES)
    line = rng.choice([1, 1, 10, 20, 30])
    meta = {"depth": rng.randrange(1, 7)} if depth_ok and rng.random() < 0.8 else {}
    return dict(node_type=rng.choice(["function", "class", "method"]), name=name, qualified_name=rng.choice([name, "app." + name]),
                file_path=rng.choice(FILES), start_line=line, end_line=line + rng.randrange(0, 40),
                signature=rng.choice([None, f"def {name}()"]), docstring=rng.choice([None, "Doc."]), summary=rng.choice([None, "Sums it up"]),
                metadata=meta)


def _hit(rng):
    name = rng.choice(NAMES)
    line = rng.choice([1, 1, 10, 20, 30])
    n = rng.choice([0, 10, 51, 60, 100, 101, 150, 1999, 2000, 2500, 2999, 3000, 4000])
    return dict(score=rng.choice([0.5, 0.75, round(rng.uniform(0.05, 0.99), 4)]), file_path=rng.choice(FILES),
                entity_type=rng.choice(["function", "class", "method", "file"]), entity_name=name, language=rng.choice(["python", "typescript"]),
                content=("x" * n) if n else rng.choice([None, ""]), start_line=line, end_line=line + 5,
                graph_node_id=rng.choice([None, name, "app." + name]), summary=rng.choice([None, "S"]))


def scenario(i):
    rng = random.Random(SEED * 1000 + i)
    graph = {role: [_node(rng, role in ("callers", "callees")) for _ in range(rng.choice([0, 0, 1, 2, 4]))] for role in ROLES}
    n_hits = rng.choice([0, 1, 3, 8, 20, 35, 70])
    keys = [n for n in NAMES if n] + ["app." + n for n in NAMES if n]
    cent = {k: {"in_degree": (d := rng.randrange(0, 80)), "out_degree": (o := rng.randrange(0, 80)), "total_degree": d + o, "relationship_count": d + o}
            for k in rng.sample(keys, rng.randrange(0, 8))}
    return dict(name=f"random_{i}", intent=rng.choice(INTENTS), entities=rng.sample([n for n in NAMES if n] + ["USERREPOSITORY", "pars"], rng.randrange(0, 4)),
                graph=graph, vector=[_hit(rng) for _ in range(n_hits)], centrality=rng.choice([cent, cent, None]))


def scenarios():
    return [scenario(i) for i in range(N_SCENARIOS)]


# ---- vector-only scenarios (no graph context): what the DEVICE re-rank (crh_rerank_vector) decides.  Expected outputs in
# tests/golden/ranking_vector_only_reference.json come from the reference's own ranker (tests/golden/gen_vector_only_goldens.py).
N_VECTOR_ONLY, VSEED = 120, 20261004
VNAMES = NAMES + ["UserRepository.save", "Parser.parse_file", "parse_file", "a_very_long_entity_name_that_is_still_below_the_sixty_four_bytes",
                  "Config.load", "löwe_ß", "do", "Helper", "helper"]


def _vhit(rng, score, files):
    name = rng.choice(VNAMES)
    line = rng.choice([1, 1, 10, 20, 30])
    n = rng.choice([0, 10, 50, 51, 60, 100, 101, 150, 1999, 2000, 2500, 2999, 3000, 4000])
    # a hit without a name is never looked up by the engine (engine.py:358-360): it carries no graph node id here either
    gid = rng.choice([None, name, "app." + name]) if name else None
    return dict(score=score, file_path=rng.choice(files), entity_type=rng.choice(["function", "class", "method", "file"]), entity_name=name,
                language=rng.choice(["python", "typescript"]), content=("x" * n) if n else rng.choice([None, ""]), start_line=line,
                end_line=line + 5, graph_node_id=gid, summary=rng.choice([None, "S"]))


def vector_only_scenario(i):
    """Hit lists as a store returns them (descending score, exact ties included), 1..128 hits, small pools of names / files /
    lines so merge keys collide (-> "hybrid"), files repeat past the per-file cap, and the total passes the cap of 50."""
    rng = random.Random(VSEED * 1000 + i)
    n_hits = rng.choice([1, 2, 5, 12, 20, 37, 64, 100, 128])
    # multiples of 2^-12: exactly representable in f32, so the device (f32 hit scores) and the reference (Python floats) start
    # from the same numbers and every f64 result can be compared to the last bit
    scores = sorted((rng.choice([0.5, 0.75, rng.randrange(-800, 4056) / 4096.0]) for _ in range(n_hits)), reverse=True)
    keys = [n for n in VNAMES if n] + ["app." + n for n in VNAMES if n]
    cent = {k: {"in_degree": (d := rng.randrange(0, 80)), "out_degree": (o := rng.randrange(0, 80)), "total_degree": d + o, "relationship_count": d + o}
            for k in rng.sample(keys, rng.randrange(0, 9))}
    ents = rng.sample([n for n in VNAMES if n] + ["USERREPOSITORY", "pars", "Parse", "zzz", "", "LÖWE_ß"], rng.randrange(0, 5))
    files = FILES if rng.random() < 0.5 else FILES + [f"pkg/m{j}.py" for j in range(17)]      # 7 files x 5 = 35 < the total cap of 50 < 24 x 5
    return dict(name=f"vector_only_{i}", intent=rng.choice(INTENTS), entities=ents, graph={},
                vector=[_vhit(rng, s, files) for s in scores], centrality=rng.choice([cent, cent, cent, None]))
