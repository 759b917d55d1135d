"""Regenerates the LegionTuning table of INTEGRATION.md from include/legion_hip.h (the header is the one place where a switch, its
environment variable, its default and its meaning are written down):
    python tools/tuning_table.py            # rewrites the block between <!-- tuning-table-begin --> and <!-- tuning-table-end -->
    python tools/tuning_table.py --check    # exit 1 if INTEGRATION.md is out of date (tests/test_tuning_cpu.py)"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table():
    hdr = open(os.path.join(ROOT, "include", "legion_hip.h")).read()
    body = hdr[hdr.index("typedef struct LegionTuning {"):hdr.index("} LegionTuning;")]
    rows = ["  | field | environment (default) | meaning |", "  |---|---|---|"]
    for m in re.finditer(r"\b(?:int32_t|uint64_t)\s+(\w+)(?:\[\d+\])?\s*;\s*/\*(.*?)\*/", body, flags=re.S):
        name, text = m.group(1), " ".join(m.group(2).split())
        env, rest = text, ""
        k = text.find(":")
        if k > 0:
            env, rest = text[:k].strip(), text[k + 1:].strip()
        rows.append("  | `{}` | `{}` | {} |".format(name, env.replace("|", "\\|"), rest.replace("|", "\\|")))
    return "\n".join(rows)


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    text = open(path).read()
    a, b = text.index("<!-- tuning-table-begin -->"), text.index("<!-- tuning-table-end -->")
    new = text[:a] + "<!-- tuning-table-begin -->\n" + table() + "\n  " + text[b:]
    if "--check" in sys.argv:
        if new != text:
            print("INTEGRATION.md: the LegionTuning table is out of date; run python tools/tuning_table.py")
            return 1
        return 0
    open(path, "w").write(new)
    return 0


if __name__ == "__main__":
    sys.exit(main())
