"""Every figure of the round's results tables comes from a committed measurement file (tools/tables_from_profiles.py builds the tables
from profiles/round6/); this test regenerates them and fails when profiles/round6/TABLES.md, or the marked copy of it in DESIGN.md /
BASELINE.md, differs -- and checks the figures quoted in prose (profiles/round6/quoted.json: text, file, key, value) to 1 %.
Round-4 review: "docs quote better numbers than the committed evidence"."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
ROUND = "round6"


def test_tables_equal_the_committed_files():
    import tables_from_profiles as tp
    path = os.path.join(ROOT, "profiles", ROUND, "TABLES.md")
    assert os.path.exists(path), "profiles/%s/TABLES.md is missing: python tools/tables_from_profiles.py %s --write" % (ROUND, ROUND)
    built = tp.build(ROUND)
    assert open(path).read() == built, "profiles/%s/TABLES.md is not what the measurement files give: re-run tools/tables_from_profiles.py %s --write" % (ROUND, ROUND)
    for doc in ("DESIGN.md", "BASELINE.md"):
        block = tp.marked(doc, ROUND)
        assert block is not None, "%s has no <!-- tables:%s --> block" % (doc, ROUND)
        assert block == built, "%s: the results table differs from profiles/%s/TABLES.md" % (doc, ROUND)


def test_quoted_figures_match_their_files():
    import tables_from_profiles as tp
    res = tp.quoted_check(ROUND)
    assert res, "profiles/%s/quoted.json is empty" % ROUND
    bad = ["%s: %r -> %s:%s = %r" % (e["doc"], e["text"], e["file"], e["path"], actual) for e, actual, ok in res if not ok]
    assert not bad, "quoted figures that are not in their document or not within 1 % of their file:\n" + "\n".join(bad)
