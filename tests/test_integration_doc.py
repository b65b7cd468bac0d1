"""INTEGRATION.md's Rust code blocks are excerpts: every line of a block that opens with `// excerpt: <path>` (other than the
`// ...` elisions) must stand verbatim in that file, so the document cannot drift from rust/ (it had)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _norm(line):
    return " ".join(line.split())


def test_rust_excerpts_are_verbatim():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```rust\n(.*?)```", doc, flags=re.S)
    checked = 0
    for b in blocks:
        lines = b.splitlines()
        m = re.match(r"// excerpt: (\S+)$", lines[0].strip())
        assert m, f"rust block without an excerpt header: {lines[0]!r}"
        src = {_norm(l) for l in open(os.path.join(ROOT, m.group(1))).read().splitlines()}
        for l in lines[1:]:
            if not l.strip() or l.strip() == "// ...":
                continue
            assert _norm(l) in src, f"{m.group(1)} has no line {l.strip()!r}"
            checked += 1
    assert len(blocks) >= 3 and checked >= 40


def test_noalloc_docs_do_not_claim_graph_capture():
    for name in ("INTEGRATION.md", "DESIGN.md", os.path.join("include", "pbrt_mi355.h")):
        text = open(os.path.join(ROOT, name)).read()
        i = text.find("pm_trace_closest_noalloc")
        assert i >= 0
        for m in re.finditer(r"\bgraphs?\b", text):
            ctx = text[max(0, m.start() - 200):m.end() + 60].lower()
            assert "not" in ctx or "cannot" in ctx, f"{name}: graph-capture claim near {ctx!r}"
