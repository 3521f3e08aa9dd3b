#!/usr/bin/env python3
"""tests/golden/ranking_vector_only_reference.json: the REFERENCE's HybridRanker (src/lattice/query/ranking/*, loaded by path
exactly as gen_goldens.py does) run on the seeded vector-only scenarios of tests/ranking_cases.py.  Survey-container-only
(needs /root/reference); writes DATA only -- per result: entity_name, file_path, start_line, source, final_score and the
four vector signals -- the quantities the device re-rank decides.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_vector_only_goldens.py
"""
import json
import sys
from pathlib import Path

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent))
import gen_goldens  # noqa: E402
import ranking_cases  # noqa: E402

SIGNALS = ("vector_similarity", "query_entity_match", "centrality", "code_quality")


def main():
    R = gen_goldens.load_reference()
    cases = []
    for i in range(ranking_cases.N_VECTOR_ONLY):
        sc = ranking_cases.vector_only_scenario(i)
        exp = gen_goldens.run_ranking(R, sc)["ranked"]
        cases.append({"name": sc["name"], "n": len(exp),
                      "rows": [[r["entity_name"], r["file_path"], r["start_line"], str(getattr(r["source"], "value", r["source"])), r["final_score"]]
                               + [r["signal_scores"].get(s) for s in SIGNALS] for r in exp]})
    out = {"generator": "tests/golden/gen_vector_only_goldens.py + tests/ranking_cases.py:vector_only_scenario", "seed": ranking_cases.VSEED,
           "reference": "src/lattice/query/ranking/{models,scorer,ranker}.py", "columns": ["entity_name", "file_path", "start_line", "source",
                                                                                            "final_score", *SIGNALS], "cases": cases}
    (HERE / "ranking_vector_only_reference.json").write_text(json.dumps(out, sort_keys=True, separators=(",", ":")))
    srcs = sorted({row[3] for c in cases for row in c["rows"]})
    print(f"wrote {len(cases)} cases; sources seen: {srcs}; sizes: {sorted({c['n'] for c in cases})}")


if __name__ == "__main__":
    main()
