"""DESIGN.md's measurement tables are generated from profiles/ (tools/make_design_tables.py): this fails when the block in
DESIGN.md differs from what the generator produces for the committed profiles -- a row cannot disagree with its profile."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_design_tables_are_the_generators_output():
    import make_design_tables as gen
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"<!-- BEGIN GENERATED: tools/make_design_tables\.py --tag (\w+).*?<!-- END GENERATED -->", text, re.S)
    assert m, "DESIGN.md has no generated measurement block"
    tag = m.group(1)
    assert os.path.exists(os.path.join(ROOT, "profiles", f"{tag}_bench.json")), f"profiles/{tag}_bench.json is not committed"
    assert m.group(0) == gen.generate(tag), "run: python tools/make_design_tables.py --tag %s --write" % tag
